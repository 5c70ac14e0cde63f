"""Is the token-major weight-gradient GEMM waiting for memory? (build: python tools/build_variant.py tndbg --only gemm_big.hip
-DTN_DBG=1; run on the GPU box with PLBERT_HIP_LIB=plbert_amd/build/ab/lib_tndbg.so)
Prints, per shape, the kernel's own stamps (time per K-tile, share of the loop spent in the vmcnt wait) for a COLD run
(operands far larger than the 256 MB Infinity Cache) and for a small problem replayed from the caches."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402

L = _lib.lib()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(Mtot, N, K, splits, reps):
    A = torch.randn(Mtot, N, device="cuda").to(torch.bfloat16)
    B = torch.randn(Mtot, K, device="cuda").to(torch.bfloat16)
    rps = (-(-Mtot // splits) + 63) // 64 * 64
    splits = -(-Mtot // rps)
    slab = torch.empty(splits, N, K, device="cuda")
    p = _lib.PlbGemmTN()
    p.A, p.lda, p.Ncols, p.B, p.ldb = A.data_ptr(), N, N, B.data_ptr(), K
    p.Mtot, p.N, p.K, p.rows_per_split, p.splits, p.slab = Mtot, N, K, rps, splits, slab.data_ptr()
    print(f"=== Mtot {Mtot} N {N} K {K} splits {splits} ({(N // 256) * (K // 256) * splits} workgroups), {reps} launches; operands "
          f"{(A.numel() + B.numel()) * 2 / 1e6:.0f} MB", flush=True)
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.plb_launch_gemm_tn_big(C.byref(p), S()) == 0
        e1.record()
        torch.cuda.synchronize()
        print(f"launch {i}: {e0.elapsed_time(e1) * 1e3:.1f} us", flush=True)


run(196608, 2304, 768, 9, 2)      # the QKV weight gradient of the step: 1.2 GB of operands, cold
run(196608, 768, 768, 28, 2)      # dense
run(24576, 2304, 768, 9, 4)       # 151 MB: replayed from the Infinity Cache
run(12288, 2304, 768, 9, 4)       # 75 MB
