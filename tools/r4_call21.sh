#!/bin/bash
# round 4, call 21: upper bound of publishing V early: the consumers read the next chunk's first fragments before the chunk barrier (racy timing ablation)
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=8 timeout -k 10 300 python tools/conv_ab.py base=$P/liblfsr_hip.so a2048=_diag/liblfsr_w4_a2048.so > gpurun_out/r4/c21_conv_ab.log 2>&1 || { tail -20 gpurun_out/r4/c21_conv_ab.log; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r4/c21_conv_ab.log
