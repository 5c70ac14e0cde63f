#!/bin/bash
# fp8 work on ONE box: the fp8 tests, then bf16 vs fp8 bench lines (fp8 with and without the 1-byte weight-gradient GEMM).
O=gpurun_out/$1
mkdir -p $O
if [ "$2" != "nopytest" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_engine.py -x -q > $O/pytest_fp8.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest_fp8.log
[ $rc -eq 124 ] && exit 1
fi
run() { name=$1; shift
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-traffic --no-staged --no-secondary "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name failed"; tail -5 $O/bench_$name.err; exit 1; }
  python - $O/bench_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["dtype"], d["ms_per_step"], "ms  loss", d["step_loss"], "timeouts", d["ln_exchange_timeouts"])
print("   ", d["roofline"]["kernel_ms_per_step"])
PY
}
run bf16 --dtype bf16
run fp8 --dtype fp8
PLBERT_FP8_TN=0 run fp8_tn16 --dtype fp8
run bf16_b --dtype bf16
run fp8_b --dtype fp8
run large_fp8 --dtype fp8 --model large
run large_bf16 --dtype bf16 --model large
