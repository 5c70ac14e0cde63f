"""Calibration only (nothing here is used by the product): the vendor BLAS that PyTorch-ROCm dispatches to
(hipBLASLt / rocBLAS) on the GEMM shapes of the step, beside this repo's kernels through the C ABI."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from plbert_amd import _lib  # noqa: E402
import gemm_bench  # noqa: E402


def time_torch(M, N, K, bias=True, iters=30):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    f = (lambda: torch.nn.functional.linear(A, W, b)) if bias else (lambda: A @ W.T)
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    L = _lib.lib()
    L.plb_set_gemm_nt_tile(0)
    L.plb_set_gemm_nt_prefetch(-1)
    for (M, N, K) in [(16384, 768, 768), (16384, 2304, 768), (16384, 2048, 768), (16384, 768, 2048), (16384, 768, 2304),
                      (8192, 8192, 8192), (4096, 4096, 4096)]:
        res = []
        for rep in range(5):
            res.append((time_torch(M, N, K), gemm_bench.time_nt(L, M, N, K, 0)[0]))
        t = sorted(r[0] for r in res)[2]
        m = sorted(r[1] for r in res)[2]
        fl = 2.0 * M * N * K
        print(f"M {M:6d} N {N:5d} K {K:5d}   torch(bias) {t*1e3:8.1f} us {fl/t/1e9:7.1f} TFLOP/s    this repo {m*1e3:8.1f} us {fl/m/1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__" and "--tn" not in sys.argv:
    main()


def tn_calibration():
    """dW = dY^T X over 196608 token rows (the batched weight gradient): vendor BLAS vs the TN pipeline kernel."""
    L = _lib.lib()
    for (Mtot, N, K) in [(196608, 2304, 768), (196608, 768, 768), (196608, 2048, 768), (196608, 768, 2048)]:
        dY = torch.randn(Mtot, N, device="cuda").to(torch.bfloat16)
        X = torch.randn(Mtot, K, device="cuda").to(torch.bfloat16)
        ts = []
        for rep in range(5):
            for _ in range(2):
                dY.T @ X
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dY.T @ X
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        t = sorted(ts)[2]
        m = sorted(gemm_bench.time_tn(L, Mtot, N, K, 1, iters=3)[0] for _ in range(5))[2]
        fl = 2.0 * Mtot * N * K
        print(f"TN Mtot {Mtot} N {N:5d} K {K:5d}   torch {t*1e3:8.1f} us {fl/t/1e9:7.1f} TFLOP/s (bf16 out)    this repo {m*1e3:8.1f} us "
              f"{fl/m/1e9:7.1f} TFLOP/s (fp32 slabs, before the reduce)", flush=True)


if __name__ == "__main__" and "--tn" in sys.argv:
    tn_calibration()
