#!/bin/bash
# Snapshot the CURRENT sources as an A/B library: plbert_amd/build/ab/lib_<name>.so, linked -Bsymbolic so its
# internal calls bind to itself when it is loaded beside the product build (tools/gemm_bench.py --libs,
# tools/ln_bench.py --libs). Extra arguments are compile flags for every source (e.g. -DNT_VAR=1).
set -e
cd "$(dirname "$0")/.."
P=plbert_amd
O=$P/build/ab/$1
mkdir -p $O
for f in gemm.hip gemm_big.hip gemm_fp8.hip gemm_ln.hip attn.hip attn_bwd_fused.hip rowops.hip mask.hip engine.cpp; do
  X=""; [ "$f" == "attn_bwd_fused.hip" ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"   # plbert_amd/build.py: EXTRA_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $X "${@:2}" -x hip -c $P/csrc/$f -o $O/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $P/build/ab/lib_$1.so $O/*.o
echo built $P/build/ab/lib_$1.so
