#!/bin/bash
# The round's closing measurements on ONE box (so the numbers compare): full GPU suite, smoke, the bench line with its
# rocprofv3 / PMC evidence, and the secondary workloads. Results under gpurun_out/final_<tag>/.
tag=$1
O=gpurun_out/final_$tag
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $tag > $O/profile.log 2>&1; echo "profile rc=$?"
bash tools/profile_round.sh ${tag}_fp8 --dtype fp8 --no-cpu-baseline --no-secondary > $O/profile_fp8.log 2>&1; echo "profile fp8 rc=$?"
run() { name=$1; shift; timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-traffic --no-secondary "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; python - "$O/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   ", d["ms_per_step"], "ms", d["value"], "tok/s", "frac", d["step_mfma_frac_wall"], "loss", d["step_loss"], "staged", (d.get("staged") or {}).get("ms_per_step"), "comm", {k: v for k, v in d["comm"].items() if k != "calibration_ms_per_step"})
PY
}
run fp8 --dtype fp8
run large --model large
run large_fp8 --model large --dtype fp8
run dual64k --num-tokens 64000
run batch96 --batch 96
run rccl_world1 --force-dist
run gpus2_shared --gpus 2 --no-roofline
