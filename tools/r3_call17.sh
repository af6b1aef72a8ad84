#!/bin/bash
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py -x -q -m gpu 2>&1 | tail -2
python tools/ffn_time.py > gpurun_out/r3/c17_ffn_abl.log 2>&1
for t in a1 a3 a4 a12 a15 a16 a31; do LFSR_HIP_LIB=$PWD/_diag/liblfsr_ffn_b3_$t.so python tools/ffn_time.py >> gpurun_out/r3/c17_ffn_abl.log 2>&1; done
python tools/ffn_time.py >> gpurun_out/r3/c17_ffn_abl.log 2>&1
grep -v amdgpu.ids gpurun_out/r3/c17_ffn_abl.log
