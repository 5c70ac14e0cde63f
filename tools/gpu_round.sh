#!/bin/bash
# One GPU-box call: parity tests, bench, rocprofv3 kernel stats of the same bench command.
set -o pipefail
mkdir -p gpurun_out
R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json
if [ "$1" == "prof" ]; then
  export TMPDIR=/tmp
  cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?"
  cd $R; ls gpurun_out/prof | head; find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -25
fi
