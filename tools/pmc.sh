#!/bin/bash
# usage: tools/pmc.sh <outname> <counters...> -- <script.py> <args...>   (run on the GPU box, from the repo root;
# the script path is taken relative to the repo root: rocprofv3 itself runs from /tmp)
name=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
R=$PWD
prog=$1; shift
case "$prog" in /*) ;; *) prog="$R/$prog" ;; esac
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
cd /tmp && timeout -k 5 150 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o $name -- python3 "$prog" "$@" > $R/gpurun_out/pmc/$name.log 2>&1
echo "rocprof rc=$?"
cd $R
python3 - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/pmc/${name}_counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    print(k, {c: f"{v:.3e}" for c, v in d.items()})
PY
