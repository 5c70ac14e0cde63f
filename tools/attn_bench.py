"""Micro-benchmark of the attention kernels through the C ABI: python tools/attn_bench.py [--libs a.so,b.so]
   fwd / bwd (dq + dkv, timed together and the dq kernel alone is not separable through the ABI)."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402


def bench(L, B, S, NH, iters=30):
    dev = "cuda"
    H = NH * 64
    T = B * S
    qkv = torch.randn(T, 3 * H, device=dev).to(torch.bfloat16)
    ctx = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
    dctx = torch.randn(T, H, device=dev).to(torch.bfloat16)
    dqkv = torch.empty(T, 3 * H, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * NH * S, device=dev)
    delta = torch.empty(B * NH * S, device=dev)
    p = _lib.PlbAttn()
    p.qkv, p.ldqkv, p.lengths, p.B, p.S, p.NH, p.H, p.scale = qkv.data_ptr(), 3 * H, None, B, S, NH, H, 0.125
    p.ctx, p.ldctx, p.lse = ctx.data_ptr(), H, lse.data_ptr()
    p.dctx, p.lddctx, p.delta, p.dqkv, p.lddqkv = dctx.data_ptr(), H, delta.data_ptr(), dqkv.data_ptr(), 3 * H
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    L.plb_profile_enable.argtypes = [C.c_int]
    L.plb_profile_read.argtypes = [C.c_void_p] * 4
    for fn in (L.plb_launch_attn_fwd, L.plb_launch_attn_bwd):
        for _ in range(3):
            assert fn(C.byref(p), s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(C.byref(p), s)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    # split of the backward into its two kernels: the library's own per-launch HIP events
    n = L.plb_profile_num_classes()
    ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
    fl, by = (C.c_double * n)(), (C.c_double * n)()
    L.plb_profile_enable(1)
    for _ in range(10):
        L.plb_launch_attn_bwd(C.byref(p), s)
    torch.cuda.synchronize()
    L.plb_profile_read(ms, cnt, fl, by)
    L.plb_profile_enable(0)
    L.plb_profile_class_name.restype = C.c_char_p
    names = [L.plb_profile_class_name(i).decode() for i in range(n)]
    split = {names[i]: ms[i] / cnt[i] * 1e3 for i in range(n) if cnt[i]}
    out.append(split.get("attn_bwd_dq", 0.0))
    out.append(split.get("attn_bwd_dkv", 0.0) + split.get("attn_bwd", 0.0))   # single-kernel form: all under "attn_bwd"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fused", action="store_true", help="time the single-kernel backward instead of dq + dkv")
    ap.add_argument("--shapes", default="32x512x12", help="comma list of BxSxNH")
    args = ap.parse_args()
    libs = [("main", _lib.lib())]
    _lib.lib().plb_set_attn_bwd_fused(1 if args.fused else 0)
    for path in [q for q in args.libs.split(",") if q]:
        libs.append((os.path.basename(path), C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)))
        libs[-1][1].plb_set_attn_bwd_fused(1 if args.fused else 0)
    res = {}
    for rep in range(args.reps):
        for (B, S, NH) in [tuple(int(x) for x in sh.split("x")) for sh in args.shapes.split(",")]:
            for name, L in libs:
                f, b, dq, dkv = bench(L, B, S, NH)
                res.setdefault((name, B, S, NH), []).append((f, b, dq, dkv))
    for (name, B, S, NH), v in res.items():
        f = sorted(x[0] for x in v)[len(v) // 2]
        b = sorted(x[1] for x in v)[len(v) // 2]
        dq = sorted(x[2] for x in v)[len(v) // 2]
        dkv = sorted(x[3] for x in v)[len(v) // 2]
        unit = B * NH * S * S * 64.0
        print(f"{name:14s} fwd {f:6.2f} us ({4*unit/f/1e6:5.0f} TF)  bwd {b:7.2f} us ({8*unit/b/1e6:5.0f} TF credited) = dq {dq:6.2f} + dkv {dkv:6.2f}", flush=True)


if __name__ == "__main__":
    main()
