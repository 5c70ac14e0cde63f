#!/bin/bash
# Timing-experiment builds of the NT pipeline GEMM: libplbert_dbgN.so = the product library with
# gemm_big.hip compiled -DNT_DBG=N (bit mask: 1 no MFMA, 2 no fragment reads, 4 no DMA after the prologue,
# 8 no K-loop barriers, 16 print the K loop's clock, 32 print phase stamps: tools/nt_stamps.py, 64 LayerNorm-backward form without its dgamma / dbeta / bias partial outputs: they cost 3.3 us of a launch). Run after the normal build; use with PLBERT_HIP_LIB=... tools/gemm_bench.py.
set -e
cd "$(dirname "$0")/.."
P=plbert_amd
mkdir -p $P/build/dbg
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNT_DBG=$n -x hip -c $P/csrc/gemm_big.hip -o $P/build/dbg/gemm_big_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNT_DBG=$n -x hip -c $P/csrc/gemm_ln.hip -o $P/build/dbg/gemm_ln_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $P/build/dbg/libplbert_dbg$n.so \
    $(ls $P/build/*.o | grep -v "/gemm_big.o\|/gemm_ln.o") $P/build/dbg/gemm_big_$n.o $P/build/dbg/gemm_ln_$n.o
  echo built $P/build/dbg/libplbert_dbg$n.so
done
