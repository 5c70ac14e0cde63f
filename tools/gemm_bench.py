"""Micro-benchmark of the NT / TN GEMM kernels through the C ABI (run on the GPU box).
   python tools/gemm_bench.py [--tiles 128,256]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from plbert_amd import _lib  # noqa: E402

SHAPES = [(16384, 768, 768), (16384, 2304, 768), (16384, 2048, 768), (16384, 768, 2048), (16384, 768, 2304),
          (8192, 8192, 8192), (4096, 4096, 4096)]


def time_nt(L, M, N, K, act, iters=20):
    dev = "cuda"
    A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K, p.Mstore = A.data_ptr(), K, B.data_ptr(), K, M, N, K, M
    p.bias = bias.data_ptr()
    p.C, p.ldc, p.C2, p.ldc2 = Cb.data_ptr(), N, C2.data_ptr(), N
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert L.plb_launch_gemm_nt(C.byref(p), act, 0, s) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.plb_launch_gemm_nt(C.byref(p), act, 0, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="128,256,384,0")
    ap.add_argument("--act", type=int, default=0)
    args = ap.parse_args()
    L = _lib.lib()
    for tile in [int(t) for t in args.tiles.split(",")]:
        L.plb_set_gemm_nt_tile(tile)
        for (M, N, K) in SHAPES:
            ms, tf = time_nt(L, M, N, K, args.act)
            print(f"tile {tile:3d}  M {M:6d} N {N:5d} K {K:5d}  {ms*1e3:9.1f} us  {tf:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
