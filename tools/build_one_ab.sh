#!/bin/bash
# A/B library that differs from the product build in ONE source: tools/build_one_ab.sh <name> <source.hip> [flags...]
# compiles that source with the extra flags and links it with the product build's other objects (plbert_amd/build/*.o,
# i.e. run plbert_amd/build.py first) into plbert_amd/build/ab/lib_<name>.so (-Bsymbolic: see tools/build_ab.sh).
set -e
cd "$(dirname "$0")/.."
P=plbert_amd
name=$1; src=$2; shift 2
O=$P/build/ab/$name
mkdir -p $O
X=""; [ "$src" == "attn_bwd_fused.hip" ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"   # plbert_amd/build.py: EXTRA_FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $X "$@" -x hip -c $P/csrc/$src -o $O/${src%.*}.o
objs=$(ls $P/build/*.o | grep -v "/${src%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $P/build/ab/lib_$name.so $objs $O/${src%.*}.o
echo built $P/build/ab/lib_$name.so
