"""Phase stamps of the NT pipeline kernel (build: tools/build_dbg.sh 32; run on the GPU box):
   PLBERT_HIP_LIB=plbert_amd/build/dbg/libplbert_dbg32.so python tools/nt_stamps.py M,N,K [act]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from plbert_amd import _lib  # noqa: E402
from gpu_util import gemm_nt  # noqa: E402


def randbf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(*shape, device='cuda', generator=g) * scale).to(torch.bfloat16)


M, N, K = (int(v) for v in sys.argv[1].split(","))
act = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = _lib.lib()
A, Bw = randbf(M, K, seed=1), randbf(N, K, scale=0.05, seed=2)
bias = torch.randn(N, device="cuda")
aux = randbf(M, N, seed=3) if act in (2, 8) else None
if act in (5, 6):   # LayerNorm forms (gemm_ln.hip)
    import ctypes as C
    res, pre = randbf(M, N, seed=4), randbf(M, N, seed=5)
    y, dx = torch.empty_like(pre), torch.empty_like(pre)
    gam, bet = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    nbn = N // (384 if N % 384 == 0 else 256)
    xchg = torch.zeros(M // 128 * nbn * nbn * 256, dtype=torch.int64, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    colp = torch.empty(2 * M // 128, 3, N, device="cuda")
    g = _lib.PlbGemmNT()
    g.A, g.lda, g.B, g.ldb, g.M, g.N, g.K, g.Mstore = A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, M
    g.res, g.ldr = res.data_ptr(), N
    g.ln_gamma, g.ln_beta, g.ln_mean, g.ln_rstd, g.ln_eps = gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1e-12
    g.ln_xchg, g.ln_err = xchg.data_ptr(), err.data_ptr()
    if act == 5:
        g.bias, g.C, g.ldc, g.C2, g.ldc2 = bias.data_ptr(), pre.data_ptr(), N, y.data_ptr(), N
    else:
        g.C, g.ldc, g.aux, g.ldaux, g.colpart = dx.data_ptr(), N, pre.data_ptr(), N, colp.data_ptr()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        print("launch", i, flush=True)
        assert L.plb_launch_gemm_nt_ln(C.byref(g), act, s) == 0
        torch.cuda.synchronize()
    sys.exit(0)
if act in (7, 8):   # gelu forms on a stashed derivative (gemm_ln.hip: plb_launch_gemm_nt_gelud)
    import ctypes as C
    Cb, C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    colp = torch.empty(2 * M // 128, N, device="cuda")
    g = _lib.PlbGemmNT()
    g.A, g.lda, g.B, g.ldb, g.M, g.N, g.K, g.Mstore = A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, M
    g.bias, g.C, g.ldc, g.C2, g.ldc2 = bias.data_ptr(), Cb.data_ptr(), N, C2.data_ptr(), N
    if act == 8:
        g.aux, g.ldaux, g.colpart = aux.data_ptr(), N, colp.data_ptr()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        print("launch", i, flush=True)
        assert L.plb_launch_gemm_nt_gelud(C.byref(g), int(act == 8), s) == 0
        torch.cuda.synchronize()
    sys.exit(0)
for i in range(3):
    print("launch", i, flush=True)
    gemm_nt(A, Bw, N, bias=bias, act=act, aux=aux)
    torch.cuda.synchronize()
