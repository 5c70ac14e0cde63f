"""Build the CURRENT sources as an A/B library plbert_amd/build/ab/lib_<name>.so (linked -Bsymbolic, selected with
PLBERT_HIP_LIB=... — tools/step_ab.py). Uses plbert_amd/build.py's source list and per-source flags.
   python tools/build_variant.py <name> [--only a.hip,b.hip] [extra compile flags ...]
--only: the extra flags go to these sources only (the others are compiled as the product build compiles them)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plbert_amd import build as B  # noqa: E402


def main():
    name, rest = sys.argv[1], sys.argv[2:]
    only = None
    if rest and rest[0] == "--only":
        only, rest = set(rest[1].split(",")), rest[2:]
    out = os.path.join(B.HERE, "build", "ab", name)
    os.makedirs(out, exist_ok=True)
    procs = []
    for src in B.SOURCES:
        extra = rest if (only is None or src in only) else []
        obj = os.path.join(out, os.path.splitext(src)[0] + ".o")
        cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        o, _ = p.communicate()
        if p.returncode:
            raise SystemExit(f"{src}:\n{o}")
        objs.append(obj)
    lib = os.path.join(B.HERE, "build", "ab", f"lib_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib, *objs], check=True)
    print("built", lib)


if __name__ == "__main__":
    main()
