"""Two-stream overlap probe: an MFMA-bound GEMM on stream A beside an HBM-bound LayerNorm on stream B.
Prints the time of N launches of each alone and of both issued together."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plbert_amd import _lib

L = _lib.lib()
dev = "cuda"
H, K = 768, 768


def gemm_args(M, N):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(max(N, 128), K, device=dev).to(torch.bfloat16)
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb = A.data_ptr(), K, B.data_ptr(), K
    p.M, p.N, p.K, p.Mstore = M, N, K, M
    p.C, p.ldc = Cb.data_ptr(), N
    return p, (A, B, Cb)


def ln_args(T):
    x = torch.randn(T, H, device=dev).to(torch.bfloat16)
    y = torch.zeros_like(x)
    g = torch.ones(H, device=dev)
    b = torch.zeros(H, device=dev)
    mean = torch.zeros(T, device=dev)
    rstd = torch.zeros(T, device=dev)
    p = _lib.PlbLayerNorm()
    p.x, p.ldx, p.gamma, p.beta, p.eps = x.data_ptr(), H, g.data_ptr(), b.data_ptr(), 1e-12
    p.y, p.ldy, p.mean, p.rstd, p.T, p.H, p.Tzero = y.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), T, H, T
    return p, (x, y, g, b, mean, rstd)


def run(fnA, fnB, n=200):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
    res = {}
    for name, fa, fb in (("A alone", fnA, None), ("B alone", None, fnB), ("A || B", fnA, fnB)):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                if fa: fa(pa)
                if fb: fb(pb)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e6
    return res


for M, N in ((16384, 768), (8192, 768), (8192, 2304), (16384, 2304)):
    g, keep1 = gemm_args(M, N)
    ln, keep2 = ln_args(M)
    r = run(lambda s: L.plb_launch_gemm_nt(C.byref(g), 0, 0, s), lambda s: L.plb_launch_ln_fwd(C.byref(ln), s))
    print(f"GEMM {M}x{N}x{K} || LN {M} rows: " + ", ".join(f"{k} {v:.1f} us" for k, v in r.items()),
          f"-> sum {r['A alone'] + r['B alone']:.1f}")
    g2, keep3 = gemm_args(M, N)
    r = run(lambda s: L.plb_launch_gemm_nt(C.byref(g), 0, 0, s), lambda s: L.plb_launch_gemm_nt(C.byref(g2), 0, 0, s))
    print(f"GEMM {M}x{N}x{K} || same GEMM: " + ", ".join(f"{k} {v:.1f} us" for k, v in r.items()),
          f"-> sum {r['A alone'] + r['B alone']:.1f}")
