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
for i in range(3):
    print("launch", i, flush=True)
    gemm_nt(A, Bw, N, bias=bias, act=act, aux=aux)
    torch.cuda.synchronize()
