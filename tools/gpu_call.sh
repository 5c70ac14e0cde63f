#!/bin/bash
# One GPU call of round 4: fp8 tests with the 256x256 gelu tiles, kernel traces (timeline) of an fp8 and a bf16 bench, bench pair.
O=$PWD/gpurun_out/$1
R=$PWD
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_engine.py -x -q > $O/pytest_fp8.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_fp8.log
[ $rc -eq 124 ] && exit 1
export TMPDIR=/tmp
B="--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-traffic --no-staged --no-secondary"
for d in fp8 bf16; do
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$d -o t -- python3 $R/bench.py $B --dtype $d > $O/trace_$d.log 2>&1; echo "trace $d rc=$?"
  cd $R
  f=$(find $O/trace_$d -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py $f --tail-ms 2.2 > $O/timeline_$d.txt 2>&1
  head -5 $O/timeline_$d.txt
  find $O/trace_$d -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/kernel_stats_$d.csv
done
bash tools/fp8_round.sh $1 nopytest
