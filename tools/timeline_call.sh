#!/bin/bash
# One GPU call: rocprofv3 kernel trace of a short bench run, then tools/timeline.py on the last step.
# usage: tools/timeline_call.sh <out-tag> [bench args...]
tag=$1; shift
R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
B="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-traffic --no-staged --no-secondary"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py $B "$@" > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f --tail-ms 2.2 > $O/timeline.txt 2>&1
rm -rf $O/trace
tail -60 $O/timeline.txt
