#!/bin/bash
# round 4, call 34: where the consumers of a half tile join barrier A
set -e
mkdir -p gpurun_out/r4
AB_ROUNDS=8 AB_GEOMS="800:n,200:n,200:y,25:n" timeout -k 10 300 python tools/conv_ab.py hag12=_diag/liblfsr_w4_hag12.so hag4=_diag/liblfsr_w4_hag4.so hag8=_diag/liblfsr_w4_hag8.so hag16=_diag/liblfsr_w4_hag16.so > gpurun_out/r4/c34_ab.log 2>&1 || { tail -20 gpurun_out/r4/c34_ab.log; exit 1; }
cat gpurun_out/r4/c34_ab.log
