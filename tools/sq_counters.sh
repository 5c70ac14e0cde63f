#!/bin/bash
# Per-kernel SQ counters of the bench step (separate rocprofv3 --pmc passes, kernel trace only beside them), summed per
# kernel name into gpurun_out/sq_<tag>.csv: MFMA-pipe busy cycles, VALU issue, LDS bank conflicts against busy cycles.
tag=$1; shift
R=$PWD
O=$R/gpurun_out/sq_$tag
mkdir -p $O
export TMPDIR=/tmp
B="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-traffic --no-staged --no-secondary $*"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  cd /tmp && timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o t -- python3 $R/bench.py $B > $O/p$i.log 2>&1; echo "pass $i ($set) rc=$?"
done
cd $R
python3 - "$O" <<'PY'
import csv, glob, collections, sys, os, re
O = sys.argv[1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return n.split("(")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], f) not in seen and r["Counter_Name"] in ("SQ_BUSY_CYCLES",): calls[k] += 1; seen.add((r["Dispatch_Id"], f))
names = sorted({c for d in agg.values() for c in d})
with open(os.path.join(O, "sq_counters.csv"), "w") as out:
    w = csv.writer(out); w.writerow(["kernel", "dispatches"] + names)
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
        w.writerow([k, calls[k]] + [f"{d.get(c, 0):.4g}" for c in names])
print(open(os.path.join(O, "sq_counters.csv")).read()[:6000])
PY
