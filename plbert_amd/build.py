"""Build libplbert_hip.so (gfx950) in-tree with hipcc.  ``python -m plbert_amd.build``"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplbert_hip.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm_fp8.hip", "gemm_fp8_ln.hip", "gemm_tn_fp8.hip", "gemm_ln.hip", "attn.hip", "attn_bwd_fused.hip", "rowops.hip", "mask.hip", "engine.cpp"]
HEADERS = ["common.h", "plbert_kernels.h", "gemm_epilogue.h", "gemm_nt_pipeline.h", "attn_common.h", os.path.join("..", "..", "include", "plbert.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-source flags. attn_bwd_fused.hip is a one-wave-per-SIMD kernel with the whole 512-entry register file: by default
# hipcc then selects the AGPR form for EVERY MFMA, so the S / dP tiles that the softmax arithmetic consumes land in
# accumulator registers and cost a v_accvgpr_read each (+50 % VALU in a loop whose VALU and MFMA time are level). With
# the VGPR form selected first, the register allocator places each MFMA result where its users want it.
# The fp8 translation units: LLVM's MachineSink pass moves the block-scaled fp8 MFMAs (8-register operand tuples) out of
# their hand-placed slots — whole phases of 16-23 MFMAs ended up back to back in the loop latch, the LDS reads and DMA
# issues that were dealt into their shadows before them (the bf16 builds keep every MFMA in its slot). With the pass off the
# K loops are as written and the kernels need 15-30 fewer registers (tools: count v_mfma runs between sched_barriers).
_NO_SINK = ["-mllvm", "-disable-machine-sink"]
EXTRA_FLAGS = {"attn_bwd_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "gemm_fp8.hip": _NO_SINK, "gemm_fp8_ln.hip": _NO_SINK, "gemm_tn_fp8.hip": _NO_SINK}


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


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble_code_objects(lib, workdir):
    """(path, disassembly) of every gfx950 code object in ``lib`` (llvm-objdump ships with ROCm; --offloading writes the
    bundles next to its input, so it runs on a copy under ``workdir``)."""
    import glob
    import shutil
    cp = shutil.copy(lib, workdir)
    subprocess.run([OBJDUMP, "--offloading", cp], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cos = sorted(glob.glob(cp + ".*gfx950"))
    if not cos:
        raise RuntimeError("no gfx950 code object extracted from " + lib)
    for co in cos:
        yield co, subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE,
                                 text=True).stdout


def verify_m0(lib=LIB, workdir=None):
    """Post-link check of EVERY build (also run by tests/test_cabi_and_host.py). The LDS-DMA statements (csrc/common.h:
    DMA16 / DMA4) write M0 and consume it inside one asm statement (`s_mov_b32 m0, sN; s_nop 0; global_load_lds_*`) and
    cannot usefully declare the clobber: hipcc reserves M0, ignores an "m0" clobber and only warns about it. That is safe
    exactly while no COMPILER-generated instruction keeps a value in M0 across such a statement, which depends on the
    hipcc version and on what else a kernel uses (builtin LDS-DMA, s_movrel, ...). So the disassembly is checked, kernel
    by kernel: a kernel that contains the statement's signature must contain NO other use of M0 — every write of M0 is
    the signature's, every LDS-DMA is the signature's, nothing reads M0; kernels whose DMAs are all the builtin's
    (csrc/gemm.hip) leave M0 to the compiler and are only required never to read it. Returns the number of asm DMA
    statements seen; raises RuntimeError on a violation; returns None when llvm-objdump is absent."""
    import re
    import tempfile
    if not os.path.exists(OBJDUMP):
        return None
    own = workdir is None
    tmp = tempfile.mkdtemp(prefix="plbert_m0_") if own else workdir
    n_asm = 0
    try:
        for co, text in disassemble_code_objects(lib, tmp):
            funcs, cur = {}, None
            for ln in text.splitlines():
                i = ln.split("//")[0].strip()
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", i)
                if m:
                    cur = funcs.setdefault(m.group(1), [])
                elif cur is not None and i and not i.endswith(":"):
                    cur.append(i)
            for name, ins in funcs.items():
                def dma(i):
                    return i.startswith("global_load_lds") or (i.startswith("buffer_load") and i.rstrip().endswith("lds"))
                sig = [k for k in range(len(ins) - 2) if re.match(r"s_mov_b32 m0, s\d+$", ins[k]) and ins[k + 1] == "s_nop 0"
                       and dma(ins[k + 2]) and " s[" in ins[k + 2]]  # scalar-base form: only the asm statements use it
                for i in ins:
                    if re.match(r"(s_movrel|v_movrel|ds_gws|ds_ordered|s_sendmsg|v_interp)", i):
                        raise RuntimeError(f"M0 check: {name}: instruction that reads M0: {i}")
                    if re.search(r"\bm0\b", i):  # M0 only ever as the destination of a scalar instruction
                        if not (re.match(r"s_\w+ m0, ", i) and not re.search(r"\bm0\b", i.split(",", 1)[1])):
                            raise RuntimeError(f"M0 check: {name}: M0 used as a source: {i}")
                if not sig:
                    continue
                n_asm += len(sig)
                owned = set(sig) | {k + 2 for k in sig}
                for k, i in enumerate(ins):
                    if (dma(i) or re.search(r"\bm0\b", i)) and k not in owned:
                        raise RuntimeError(f"M0 check: {name}: M0 / LDS-DMA outside the asm DMA statements: {ins[max(0, k - 2):k + 3]}")
    finally:
        if own:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return n_asm


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
    try:
        verify_m0(LIB)   # a user build with another hipcc must not ship kernels in which the compiler uses M0 around the DMAs
    except Exception:
        os.remove(LIB)
        raise
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
