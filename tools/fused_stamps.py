"""Diagnostic: per-phase shader-clock sums of the single-kernel attention backward (a library built with
-DFUSED_DBG=1024: tools/build_one_ab.sh stamps attn_bwd_fused.hip -DFUSED_DBG=1024).
   python tools/fused_stamps.py plbert_amd/build/ab/lib_stamps.so [BxSxNH]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402

_lib.lib()
L = C.CDLL(os.path.abspath(sys.argv[1]), mode=C.RTLD_LOCAL)
B, S, NH = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16x512x16").split("x"))
H, T, dev = NH * 64, B * S, "cuda"
qkv = torch.randn(T, 3 * H, device=dev).to(torch.bfloat16)
ctx = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
dctx = torch.randn(T, H, device=dev).to(torch.bfloat16)
dqkv = torch.empty(T, 3 * H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * NH * S, device=dev)
delta = torch.zeros(max(B * NH * S, B * NH * 32), device=dev)
p = _lib.PlbAttn()
p.qkv, p.ldqkv, p.lengths, p.B, p.S, p.NH, p.H, p.scale = qkv.data_ptr(), 3 * H, None, B, S, NH, H, 0.125
p.ctx, p.ldctx, p.lse = ctx.data_ptr(), H, lse.data_ptr()
p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.plb_launch_attn_fwd(C.byref(p), s)
for _ in range(5):
    L.plb_launch_attn_bwd_fused(C.byref(p), s)
torch.cuda.synchronize()
st = delta[: B * NH * 32].reshape(B * NH, 4, 8).cpu()
names = ["stage issue", "key-owner", "dQ phase+store", "stat store", "wait+barrier", "loop head", "loop total"]
med = st.median(dim=0).values
for w in range(4):
    print(f"wave {w}: " + "  ".join(f"{n} {med[w, k]:.1f}" for k, n in enumerate(names)) + "  (kilo-cycles, median over workgroups)")
