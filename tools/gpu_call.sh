#!/bin/bash
# One GPU call of round 4: stamps of the fp8 / bf16 NT forms, then the whole GPU suite.
O=gpurun_out/$1
mkdir -p $O
PLBERT_HIP_LIB=plbert_amd/build/ab/lib_dbg32.so timeout -k 10 300 python tools/nt_stamps_fp8.py > $O/stamps.txt 2>&1; echo "stamps rc=$?"
grep -c "blk" $O/stamps.txt
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
