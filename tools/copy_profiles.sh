#!/bin/bash
# Copy the evidence of a tools/final_round.sh run (gpurun_out/prof_<tag>, gpurun_out/final_<tag>) into profiles/ under
# the round's naming: profiles/<tag>_bench.json, _bench_kernel_stats.csv, _hbm_traffic_pmc.csv, _<workload>_bench.json.
tag=$1
cp gpurun_out/prof_$tag/bench.json profiles/${tag}_bench.json
cp gpurun_out/prof_$tag/kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp gpurun_out/prof_$tag/hbm_traffic_pmc.csv profiles/${tag}_hbm_traffic_pmc.csv
if [ -d gpurun_out/prof_${tag}_fp8 ]; then
  cp gpurun_out/prof_${tag}_fp8/bench.json profiles/${tag}_fp8_profiled_bench.json
  cp gpurun_out/prof_${tag}_fp8/kernel_stats.csv profiles/${tag}_fp8_bench_kernel_stats.csv
  cp gpurun_out/prof_${tag}_fp8/hbm_traffic_pmc.csv profiles/${tag}_fp8_hbm_traffic_pmc.csv
fi
for w in fp8 large large_fp8 dual64k batch96 rccl_world1 gpus2_shared; do
  [ -s gpurun_out/final_$tag/$w.json ] && cp gpurun_out/final_$tag/$w.json profiles/${tag}_${w}_bench.json
done
tail -3 gpurun_out/final_$tag/pytest_gpu.log > profiles/${tag}_pytest_gpu_tail.txt
ls profiles/${tag}_*
