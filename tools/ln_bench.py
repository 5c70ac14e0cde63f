"""Micro-benchmark of the LayerNorm kernels through the C ABI: python tools/ln_bench.py [--libs a.so,b.so]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import _lib  # noqa: E402


def bench(L, T, H, iters=50, nb=1024):
    dev = "cuda"
    x = torch.randn(T, H, device=dev).to(torch.bfloat16)
    dy = torch.randn(T, H, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
    mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
    part = torch.empty(nb, 3 * H, device=dev)
    p = _lib.PlbLayerNorm()
    p.x, p.ldx, p.gamma, p.beta, p.eps = x.data_ptr(), H, g.data_ptr(), b.data_ptr(), 1e-12
    p.y, p.ldy, p.mean, p.rstd, p.T, p.H, p.Tzero = y.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), T, H, T
    p.dy, p.lddy, p.dx, p.lddx, p.partials, p.nblocks = dy.data_ptr(), H, dx.data_ptr(), H, part.data_ptr(), nb
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for fn, bytes_ in ((L.plb_launch_ln_fwd, T * H * 4), (L.plb_launch_ln_bwd, T * H * 6)):
        for _ in range(5):
            assert fn(C.byref(p), s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(C.byref(p), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        out.append((us, bytes_ / (us * 1e-6) / 1e12))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--blocks", type=lambda v: [int(x) for x in v.split(",")], default=[1024],
                    help="partial-sum blocks of the backward (the engine uses 1024)")
    args = ap.parse_args()
    libs = [("main", _lib.lib())]
    for path in [q for q in args.libs.split(",") if q]:
        libs.append((os.path.basename(path), C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)))
    res = {}
    for rep in range(args.reps):
        for (T, H) in ((16384, 768), (8192, 1024)):
            for name, L in libs:
                for nb in args.blocks:
                    (f_us, f_tb), (b_us, b_tb) = bench(L, T, H, nb=nb)
                    res.setdefault((f"{name}/{nb}", T, H), []).append((f_us, b_us))
    for (name, T, H), v in res.items():
        f = sorted(x[0] for x in v)[len(v) // 2]
        b = sorted(x[1] for x in v)[len(v) // 2]
        print(f"{name:22s} T {T:6d} H {H:5d}  fwd {f:7.2f} us ({T*H*4/f/1e6:5.2f} TB/s)   bwd {b:7.2f} us ({T*H*6/b/1e6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
