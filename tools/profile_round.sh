#!/bin/bash
# One GPU-box call that produces the round's evidence under gpurun_out/prof_<tag>/ (copy what is to be judged into
# profiles/): the full bench line, rocprofv3 kernel stats of the same bench command, and per-kernel HBM traffic from
# PMC counters (FETCH_SIZE and WRITE_SIZE in separate passes, kernel trace only beside them — MI355X_MICROARCH.md).
# usage: tools/profile_round.sh <tag> [extra bench args...]
set -o pipefail
tag=$1; shift
R=$PWD
O=$R/gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python bench.py "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
B="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-traffic --no-staged --no-secondary"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py $B "$@" > $O/stats.log 2>&1; echo "stats rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o t -- python3 $R/bench.py $B "$@" > $O/pmc_$ctr.log 2>&1; echo "pmc $ctr rc=$?"
done
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/kernel_stats.csv
python3 - "$O" <<'PY'
import csv, glob, collections, sys, os, re
O = sys.argv[1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]
agg = collections.defaultdict(lambda: dict(n=0, fetch=0.0, write=0.0, ns=0.0, nt=0))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(O, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print("no counter file for", ctr); continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr: continue
        a = agg[short(r["Kernel_Name"])]
        if ctr == "FETCH_SIZE": a["fetch"] += float(r["Counter_Value"]); a["n"] += 1
        else: a["write"] += float(r["Counter_Value"])
    t = glob.glob(os.path.join(O, "pmc_" + ctr, "**", "*kernel_trace.csv"), recursive=True)
    if t and ctr == "FETCH_SIZE":
        for r in csv.DictReader(open(t[0])):
            a = agg[short(r["Kernel_Name"])]
            a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a["nt"] += 1
with open(os.path.join(O, "hbm_traffic_pmc.csv"), "w") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "launches", "avg_us(profiled pass)", "fetch_MB_per_launch(FETCH_SIZE*1024*2)", "write_MB_per_launch(WRITE_SIZE*1024)",
                "GB/s", "frac_of_8TBps"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        if not a["n"] or not a["nt"]: continue
        fb, wb, us = a["fetch"] / a["n"] * 2048, a["write"] / a["n"] * 1024, a["ns"] / a["nt"] / 1e3
        gbs = (fb + wb) / (us * 1e-6) / 1e9 if us else 0
        w.writerow([k, a["n"], f"{us:.2f}", f"{fb/1e6:.2f}", f"{wb/1e6:.2f}", f"{gbs:.0f}", f"{gbs/8000:.3f}"])
print(open(os.path.join(O, "hbm_traffic_pmc.csv")).read())
PY
head -c 1500 $O/bench.json; echo; head -22 $O/kernel_stats.csv | cut -c1-150
