#!/bin/bash
# One GPU call of round 4: tail-order A/B (bf16 and fp8), then engine / comm / fp8 tests.
O=$PWD/gpurun_out/$1
mkdir -p $O
python tools/step_ab.py --reps 3 side:PLBERT_TAIL_ORDER=side main:PLBERT_TAIL_ORDER=main > $O/ab_tail_bf16.txt 2>&1; tail -4 $O/ab_tail_bf16.txt
python tools/step_ab.py --reps 3 --args "--dtype fp8" side:PLBERT_TAIL_ORDER=side main:PLBERT_TAIL_ORDER=main > $O/ab_tail_fp8.txt 2>&1; tail -4 $O/ab_tail_fp8.txt
timeout -k 10 800 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fp8.py tests/test_gpu_comm_rccl.py tests/test_gpu_comm_fake_rccl.py tests/test_gpu_dual_head.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
