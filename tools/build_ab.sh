#!/bin/bash
# Snapshot the current objects as an A/B library: pl-bert_amd/build/ab/lib_<name>.so, linked -Bsymbolic so
# its internal calls bind to itself when it is loaded beside the product build (tools/gemm_bench.py --libs).
set -e
cd "$(dirname "$0")/.."
P=pl-bert_amd
mkdir -p $P/build/ab
# extra arguments are compile flags for gemm_big.hip (e.g. -DNT_VAR=1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "${@:2}" -x hip -c $P/csrc/gemm_big.hip -o $P/build/ab/gemm_big_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $P/build/ab/lib_$1.so \
  $P/build/gemm.o $P/build/ab/gemm_big_$1.o $P/build/attn.o $P/build/rowops.o $P/build/mask.o $P/build/engine.o
echo built $P/build/ab/lib_$1.so
