"""Phase stamps of the NT pipeline kernel, bf16 and fp8 forms side by side (build: python tools/build_variant.py dbg32
--only gemm_big.hip,gemm_ln.hip,gemm_fp8.hip,gemm_fp8_ln.hip -DNT_DBG=32; run on the GPU box):
   PLBERT_HIP_LIB=plbert_amd/build/ab/lib_dbg32.so python tools/nt_stamps_fp8.py
Each launch prints (from workgroup 17 of every 256) fill | K loop | epilogue phases in units of 10 ns."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402

L = _lib.lib()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def q8(x, bf8):
    dt, mx = (torch.float8_e5m2, 57344.0) if bf8 else (torch.float8_e4m3fn, 448.0)
    s = mx / float(x.abs().max())
    return (x.float() * s).clamp(-mx, mx).to(dt).view(torch.uint8), 1.0 / s


def run(M, N, K, act, fp8):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bf8 = act in (6, 8)
    p = _lib.PlbGemmNT()
    keep = []
    if fp8:
        A8, da = q8(A, bf8)
        W8, dw = q8(W, False)
        deq = torch.tensor([da, dw], device="cuda")
        p.A, p.B, p.deq_a, p.deq_b = A8.data_ptr(), W8.data_ptr(), deq.data_ptr(), deq.data_ptr() + 4
        keep += [A8, W8, deq]
    else:
        Ab, Wb = A.to(torch.bfloat16), W.to(torch.bfloat16)
        p.A, p.B = Ab.data_ptr(), Wb.data_ptr()
        keep += [Ab, Wb]
    p.lda, p.ldb, p.M, p.N, p.K, p.Mstore = K, K, M, N, K, M
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    pre = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    o1, o2 = torch.empty_like(pre), torch.empty_like(pre)
    img = torch.empty(M, N, dtype=torch.uint8, device="cuda")
    qs, amax = torch.tensor([1.0], device="cuda"), torch.zeros(1024, device="cuda")
    gam, bet = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    nbn = N // (384 if N % 384 == 0 else 256)
    xchg = torch.zeros(M // 128 * nbn * nbn * 256, dtype=torch.int64, device="cuda")
    err = torch.zeros(2, dtype=torch.int32, device="cuda")
    colp = torch.empty(2 * M // 128, 3, N, device="cuda")
    keep += [bias, res, pre, o1, o2, img, qs, amax, gam, bet, mean, rstd, xchg, err, colp]
    if fp8 and act != 0:   # (the engine's plain fp8 launches — QKV forward, dCtx, dX of application 0 — write no image)
        p.C8, p.ldc8, p.q_scale, p.q_amax, p.c8_bf8 = img.data_ptr(), N, qs.data_ptr(), amax.data_ptr(), int(bf8)
    if act in (5, 6):
        p.res, p.ldr = res.data_ptr(), N
        p.ln_gamma, p.ln_beta, p.ln_mean, p.ln_rstd, p.ln_eps = gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1e-12
        p.ln_xchg, p.ln_err = xchg.data_ptr(), err.data_ptr()
        if act == 5:
            p.bias, p.C, p.ldc, p.C2, p.ldc2 = bias.data_ptr(), o1.data_ptr(), N, o2.data_ptr(), N
        else:
            p.C, p.ldc, p.aux, p.ldaux, p.colpart = o1.data_ptr(), N, pre.data_ptr(), N, colp.data_ptr()
        fn = (lambda: L.plb_launch_gemm_nt_fp8_ln(C.byref(p), act, int(bf8), S())) if fp8 else (lambda: L.plb_launch_gemm_nt_ln(C.byref(p), act, S()))
    elif act in (7, 8):
        p.bias = bias.data_ptr() if act == 7 else None
        p.C, p.ldc = o1.data_ptr(), N
        if act == 7:
            if not fp8:
                p.C2, p.ldc2 = o2.data_ptr(), N
        else:
            p.aux, p.ldaux, p.colpart = pre.data_ptr(), N, colp.data_ptr()
            if fp8:
                p.C = None
        fn = (lambda: L.plb_launch_gemm_nt_fp8_gelud(C.byref(p), int(act == 8), int(bf8), S())) if fp8 else \
             (lambda: L.plb_launch_gemm_nt_gelud(C.byref(p), int(act == 8), S()))
    else:
        p.bias, p.C, p.ldc = bias.data_ptr(), o1.data_ptr(), N
        fn = (lambda: L.plb_launch_gemm_nt_fp8(C.byref(p), 0, int(bf8), S())) if fp8 else (lambda: L.plb_launch_gemm_nt(C.byref(p), 0, 0, S()))
    print(f"=== M {M} N {N} K {K} act {act} {'fp8' if fp8 else 'bf16'}", flush=True)
    for i in range(3):
        assert fn() == 0
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    return keep


for shape in ((16384, 768, 768, 5), (16384, 768, 2048, 5), (16384, 768, 2048, 6), (16384, 768, 2304, 6), (16384, 2048, 768, 7),
              (16384, 2048, 768, 8), (16384, 2304, 768, 0), (16384, 768, 768, 0)):
    for fp8 in (False, True):
        run(*shape, fp8)
