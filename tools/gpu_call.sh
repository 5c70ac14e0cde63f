#!/bin/bash
# One GPU call of round 4: fp8 diagnostics, then the whole GPU suite, then the bench pair.
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 300 python tools/fp8_diag.py > $O/diag_tn8.txt 2>&1; echo "diag rc=$?"; grep -- "---" $O/diag_tn8.txt | cut -c1-150
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
bash tools/fp8_round.sh $1 nopytest
