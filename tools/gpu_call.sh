#!/bin/bash
# One GPU call of round 4: fp8 epilogue conversion changes A/B (fp8), bf16 sanity A/B, then the whole GPU suite.
O=$PWD/gpurun_out/$1
mkdir -p $O
python tools/step_ab.py --reps 3 --args "--dtype fp8" prev:PLBERT_HIP_LIB=plbert_amd/build/ab/lib_prev.so new > $O/ab_f8conv_fp8.txt 2>&1; tail -4 $O/ab_f8conv_fp8.txt
python tools/step_ab.py --reps 2 prev:PLBERT_HIP_LIB=plbert_amd/build/ab/lib_prev.so new > $O/ab_f8conv_bf16.txt 2>&1; tail -4 $O/ab_f8conv_bf16.txt
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_gpu.log
