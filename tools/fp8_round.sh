#!/bin/bash
# fp8 work on ONE box: hardware probe of the transposed byte reads, the fp8 tests, bf16 vs fp8 bench lines.
O=gpurun_out/$1
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe_tr8.hip -o /tmp/probe_tr8 2>/dev/null && /tmp/probe_tr8 > $O/probe_tr8.txt 2>&1; echo "probe rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q > $O/pytest_fp8.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest_fp8.log
[ $rc -eq 124 ] && exit 1
for d in bf16 fp8; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-traffic --no-staged --no-secondary --dtype $d > $O/bench_$d.json 2> $O/bench_$d.err || { echo "bench $d failed"; tail -5 $O/bench_$d.err; exit 1; }
  python - $O/bench_$d.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["dtype"], d["ms_per_step"], "ms  loss", d["step_loss"], "timeouts", d["ln_exchange_timeouts"])
print("   ", d["roofline"]["kernel_ms_per_step"])
PY
done
