#!/bin/bash
# one GPU-box call: kernel tests then engine tests, logs under gpurun_out/
mkdir -p gpurun_out
python -m pytest ${@:-tests} -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -60 gpurun_out/pytest_gpu.log
exit $rc
