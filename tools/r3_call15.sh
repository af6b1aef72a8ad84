#!/bin/bash
set -e
mkdir -p gpurun_out/r3
python tools/lin_time.py > gpurun_out/r3/c15_lin_abl.log 2>&1
for t in a1 a2 a4 a8 a3 a12 a7 a11; do LFSR_HIP_LIB=$PWD/_diag/liblfsr_rowgemm_b3_$t.so python tools/lin_time.py >> gpurun_out/r3/c15_lin_abl.log 2>&1; done
python tools/lin_time.py >> gpurun_out/r3/c15_lin_abl.log 2>&1
cat gpurun_out/r3/c15_lin_abl.log
