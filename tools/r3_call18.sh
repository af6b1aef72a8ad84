#!/bin/bash
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_epit.py tests/test_gpu_lft.py -x -q -m gpu 2>&1 | tail -2
python tools/ffn_time.py 2>&1 | grep -v amdgpu.ids
LFSR_HIP_LIB=$PWD/_diag/liblfsr_nodot2.so python tools/ffn_time.py 2>&1 | grep -v amdgpu.ids
python tools/ffn_time.py 2>&1 | grep -v amdgpu.ids
LFSR_HIP_LIB=$PWD/_diag/liblfsr_nodot2.so python tools/ffn_time.py 2>&1 | grep -v amdgpu.ids
