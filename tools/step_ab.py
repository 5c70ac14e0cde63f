"""Step-level A/B on ONE box: runs bench.py round-robin under different environments (fresh child process each, the
parent never touches the GPU) and prints ms/step plus the kernel classes that differ.
   python tools/step_ab.py [--reps 2] [--args "--model large"] label[:ENV=VAL[,ENV=VAL...]] ...
e.g. python tools/step_ab.py base ntstore:PLBERT_HIP_LIB=plbert_amd/build/ab/lib_ntstore.so"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--args", default="")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for rep in range(a.reps):
        for v in a.variants:
            label, _, envs = v.partition(":")
            env = dict(os.environ)
            for kv in [e for e in envs.split(",") if e]:
                k, _, val = kv.partition("=")
                env[k] = os.path.abspath(os.path.join(ROOT, val)) if k == "PLBERT_HIP_LIB" else val
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "10", "--no-cpu-baseline",
                   "--no-traffic", "--no-staged", "--no-secondary"] + a.args.split()
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(label, "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            k = d.get("roofline", {}).get("kernel_ms_per_step", {})
            res.setdefault(label, []).append((d["ms_per_step"], k))
            top = {n: t for n, t in k.items() if t >= 0.4}
            print(f"{label:12s} {d['ms_per_step']:8.3f} ms/step  {top}", flush=True)
    print("--- median ms/step")
    for label, v in res.items():
        ms = sorted(x[0] for x in v)
        print(f"{label:12s} {ms[len(ms) // 2]:8.3f}  (runs: {', '.join(f'{x:.3f}' for x in ms)})")


if __name__ == "__main__":
    main()
