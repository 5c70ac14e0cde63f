#!/bin/bash
# One GPU call of round 4: fp8 vs bf16 convergence, soak, SQ counters of the fp8 step.
O=$PWD/gpurun_out/$1
mkdir -p $O
timeout -k 10 300 python tools/fp8_convergence.py 640 2e-4 > $O/fp8_convergence.txt 2>&1; echo "convergence rc=$?"; grep -v amdgpu $O/fp8_convergence.txt | tail -24
timeout -k 10 300 python tools/soak.py 1500 > $O/soak_bf16.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu $O/soak_bf16.txt | tail -5
bash tools/sq_counters.sh ${1}_fp8 --dtype fp8 > $O/sq_fp8.log 2>&1; echo "sq rc=$?"; cp gpurun_out/sq_${1}_fp8/sq_counters.csv $O/sq_counters_fp8.csv; head -14 $O/sq_counters_fp8.csv | cut -c1-220
