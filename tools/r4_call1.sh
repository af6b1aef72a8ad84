#!/bin/bash
# round 4, call 1: where the 3x3 conv's time goes on the current kernel (ablations), touch-ahead and producer-only barrier A variants
set -e
mkdir -p gpurun_out/r4
L=""
for t in a1 a32 a2 a6 a33 a39 a55 t1 t2 ps pst1 g3 g12 g33; do L="$L $t=_diag/liblfsr_w4_$t.so"; done
P=$(ls -d ntire-2026-*_amd)
timeout -k 10 500 python tools/conv_ab.py base=$P/liblfsr_hip.so $L > gpurun_out/r4/c1_conv_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r4/c1_conv_ab.log
