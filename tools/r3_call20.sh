#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_epit.py tests/test_gpu_lft.py -x -q -m gpu > gpurun_out/r3/c20_tests.log 2>&1 || { tail -40 gpurun_out/r3/c20_tests.log; exit 1; }
tail -2 gpurun_out/r3/c20_tests.log
python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
LFSR_LNLIN=0 python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
LFSR_LNLIN=0 python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
