"""A/B of LayerNorm fused into the producing GEMM's epilogue against GEMM + LayerNorm kernel, through the C ABI:
python tools/ln_fuse_bench.py   (shapes of the two model configurations; times are per pair / per fused launch)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda"
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def one(M, N, K):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias, gam, bet = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    pre, y, dx = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(3))
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    nbn = N // (384 if N % 384 == 0 else 256)
    xchg = torch.zeros(M // 128 * nbn * nbn * 256, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    colp = torch.empty(2 * M // 128, 3, N, device=dev)
    part = torch.empty(1024, 3 * N, device=dev)
    g = _lib.PlbGemmNT()
    g.A, g.lda, g.B, g.ldb, g.M, g.N, g.K, g.Mstore = A.data_ptr(), K, B.data_ptr(), K, M, N, K, M
    g.bias, g.res, g.ldr, g.C, g.ldc, g.C2, g.ldc2 = bias.data_ptr(), res.data_ptr(), N, pre.data_ptr(), N, y.data_ptr(), N
    g.ln_gamma, g.ln_beta, g.ln_mean, g.ln_rstd, g.ln_eps = gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1e-12
    g.ln_xchg, g.ln_err = xchg.data_ptr(), err.data_ptr()
    ln = _lib.PlbLayerNorm()
    ln.x, ln.ldx, ln.gamma, ln.beta, ln.eps = pre.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), 1e-12
    ln.y, ln.ldy, ln.mean, ln.rstd, ln.T, ln.H, ln.Tzero = y.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(), M, N, M
    ln.dy, ln.lddy, ln.dx, ln.lddx, ln.partials, ln.nblocks = y.data_ptr(), N, dx.data_ptr(), N, part.data_ptr(), 512

    def unf_f():
        L.plb_launch_gemm_nt(C.byref(g), 0, 0, s); L.plb_launch_ln_fwd(C.byref(ln), s)

    t_gemm = timeit(lambda: L.plb_launch_gemm_nt(C.byref(g), 0, 0, s))
    t_uf = timeit(unf_f)
    t_ff = timeit(lambda: L.plb_launch_gemm_nt_ln(C.byref(g), 5, s))
    # backward: the GEMM's output (y buffer here) is the LayerNorm's dy
    gb = _lib.PlbGemmNT.from_buffer_copy(g)
    gb.bias, gb.C, gb.C2 = None, y.data_ptr(), None

    def unf_b():
        L.plb_launch_gemm_nt(C.byref(gb), 0, 0, s); L.plb_launch_ln_bwd(C.byref(ln), s)

    t_ub = timeit(unf_b)
    gb.C, gb.aux, gb.ldaux, gb.colpart = dx.data_ptr(), pre.data_ptr(), N, colp.data_ptr()
    t_fb = timeit(lambda: L.plb_launch_gemm_nt_ln(C.byref(gb), 6, s))
    # timing experiments (wrong results): every row's pre segment from one line / no residual — what the partial-line
    # epilogue loads cost
    gb.ldaux = 0
    t_fb_noaux = timeit(lambda: L.plb_launch_gemm_nt_ln(C.byref(gb), 6, s))
    gb.ldaux = N
    gb.res = None
    t_fb_nores = timeit(lambda: L.plb_launch_gemm_nt_ln(C.byref(gb), 6, s))
    print(f"    bwd fused with the pre loads served from one line {t_fb_noaux:6.1f}, without residual {t_fb_nores:6.1f} us")
    assert int(err.item()) == 0
    print(f"M {M:6d} N {N:5d} K {K:5d}: GEMM alone {t_gemm:6.1f} | fwd GEMM+LN {t_uf:6.1f} -> fused {t_ff:6.1f} us | "
          f"bwd GEMM+LN {t_ub:6.1f} -> fused {t_fb:6.1f} us", flush=True)


for shape in ((16384, 768, 768), (16384, 768, 2048), (16384, 768, 2304), (8192, 1024, 1024), (8192, 1024, 4096), (49152, 768, 768)):
    one(*shape)
