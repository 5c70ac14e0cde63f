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


MSTORE_ZERO = False
WITH_RES = False


def time_nt(L, M, N, K, act, iters=20):
    dev = "cuda"
    A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    p = _lib.PlbGemmNT()
    p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K, p.Mstore = A.data_ptr(), K, B.data_ptr(), K, M, N, K, (0 if MSTORE_ZERO else M)
    p.bias = bias.data_ptr()
    p.C, p.ldc, p.C2, p.ldc2 = Cb.data_ptr(), N, C2.data_ptr(), N
    if WITH_RES:
        R = torch.randn(M, N, device=dev).to(torch.bfloat16)
        p.res, p.ldr = R.data_ptr(), N
    if act == 2:
        U = torch.randn(M, N, device=dev).to(torch.bfloat16)
        p.aux, p.ldaux = U.data_ptr(), N
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


TN_SHAPES = [(196608, 2304, 768), (196608, 768, 768), (196608, 2048, 768), (196608, 768, 2048),
             (16384, 2304, 768), (16384, 768, 768), (32768, 2304, 768)]


def time_tn(L, Mtot, N, K, big, iters=5):
    dev = "cuda"
    A = torch.randn(Mtot, N, device=dev).to(torch.bfloat16)
    B = torch.randn(Mtot, K, device=dev).to(torch.bfloat16)
    tiles = (N // 256) * (K // 256) if big else ((N + 127) // 128) * ((K + 127) // 128)
    splits = 8 * max(1, 32 // tiles) if big else max(1, 768 // tiles)
    rps = ((Mtot + splits - 1) // splits + 63) // 64 * 64
    splits = (Mtot + rps - 1) // rps
    slab = torch.empty(splits * N * K, dtype=torch.float32, device=dev)
    p = _lib.PlbGemmTN()
    p.A, p.lda, p.Ncols, p.B, p.ldb = A.data_ptr(), N, N, B.data_ptr(), K
    p.Mtot, p.N, p.K, p.rows_per_split, p.splits, p.slab = Mtot, N, K, rps, splits, slab.data_ptr()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = L.plb_launch_gemm_tn_big if big else L.plb_launch_gemm_tn
    assert fn(C.byref(p), s) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(C.byref(p), s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * Mtot * N * K / (ms * 1e-3) / 1e12, splits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="128,256,384,0")
    ap.add_argument("--act", type=int, default=0)
    ap.add_argument("--tn", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--shapes", default="", help="M,N,K;M,N,K;... instead of the built-in NT list")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--libs", default="", help="other builds of libplbert_hip.so to time beside the product build")
    ap.add_argument("--prefetch", default="-1", help="comma list of 0/1: NT big-tile K-loop variants to time")
    ap.add_argument("--res", action="store_true", help="add a bf16 residual operand in the epilogue")
    ap.add_argument("--nostore", action="store_true", help="Mstore = 0: the epilogue computes but stores nothing")
    args = ap.parse_args()
    global MSTORE_ZERO, WITH_RES
    MSTORE_ZERO = args.nostore
    WITH_RES = args.res
    L = _lib.lib()
    if args.tn:
        libs = [("main", L)] + [(os.path.basename(q), C.CDLL(os.path.abspath(q), mode=C.RTLD_LOCAL))
                                for q in args.libs.split(",") if q]
        res = {}
        shapes_tn = TN_SHAPES[:2] if args.quick else TN_SHAPES[4:] if args.small else TN_SHAPES
        for rep in range(args.reps):
            for (M, N, K) in shapes_tn:
                for (ln, lib) in libs:
                    ms, tf, sp = time_tn(lib, M, N, K, 1, iters=3)
                    res.setdefault((ln, M, N, K, sp), []).append(ms)
        for (ln, M, N, K, sp), v in res.items():
            v = sorted(v)
            med = v[len(v) // 2]
            print(f"{ln:14s} tn big Mtot {M} N {N:5d} K {K:5d} splits {sp:3d}  median {med*1e3:9.1f} us  "
                  f"{2.0*M*N*K/(med*1e-3)/1e12:7.1f} TFLOP/s", flush=True)
        return
    shapes = [tuple(int(v) for v in t.split(",")) for t in args.shapes.split(";")] if args.shapes else SHAPES
    # A/B protocol: box-to-box and minute-to-minute drift is 5-10 %, so variants are timed round-robin
    # inside one process (other builds of the library: --libs a.so,b.so) and the median / min over --reps
    # rounds is reported per variant.
    libs = [("main", L)]
    for path in [q for q in args.libs.split(",") if q]:
        X = C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)
        X.plb_set_gemm_nt_tile.argtypes = [C.c_int]
        X.plb_set_gemm_nt_prefetch.argtypes = [C.c_int]
        libs.append((os.path.basename(path), X))
    variants = [(ln, lib, pf, tile) for (ln, lib) in libs for pf in [int(t) for t in args.prefetch.split(",")]
                for tile in [int(t) for t in args.tiles.split(",")]]
    time_nt(L, *SHAPES[0], args.act)  # warm the clocks
    res = {}
    for rep in range(args.reps):
        for (M, N, K) in shapes:
            for (ln, lib, pf, tile) in variants:
                lib.plb_set_gemm_nt_prefetch(pf)
                lib.plb_set_gemm_nt_tile(tile)
                ms, tf = time_nt(lib, M, N, K, args.act)
                res.setdefault((ln, pf, tile, M, N, K), []).append(ms)
    for (ln, pf, tile, M, N, K), v in res.items():
        v = sorted(v)
        med = v[len(v) // 2]
        print(f"{ln:14s} pf {pf:2d} tile {tile:4d}  M {M:6d} N {N:5d} K {K:5d}  median {med*1e3:8.1f} us  min {v[0]*1e3:8.1f} us"
              f"  {2.0*M*N*K/(med*1e-3)/1e12:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
