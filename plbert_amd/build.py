"""Build libplbert_hip.so (gfx950) in-tree with hipcc.  ``python -m plbert_amd.build``"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplbert_hip.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm_fp8.hip", "gemm_ln.hip", "attn.hip", "attn_bwd_fused.hip", "rowops.hip", "mask.hip", "engine.cpp"]
HEADERS = ["common.h", "plbert_kernels.h", "gemm_epilogue.h", "gemm_nt_pipeline.h", "attn_common.h", os.path.join("..", "..", "include", "plbert.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-source flags. attn_bwd_fused.hip is a one-wave-per-SIMD kernel with the whole 512-entry register file: by default
# hipcc then selects the AGPR form for EVERY MFMA, so the S / dP tiles that the softmax arithmetic consumes land in
# accumulator registers and cost a v_accvgpr_read each (+50 % VALU in a loop whose VALU and MFMA time are level). With
# the VGPR form selected first, the register allocator places each MFMA result where its users want it.
EXTRA_FLAGS = {"attn_bwd_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library next to the package."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
