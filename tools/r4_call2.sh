#!/bin/bash
# round 4, call 2: the producers' drain moved in front of barrier A (+ epilogue operand a step ahead, branch-free activation), barrier-A position, clock probes
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
L=""
for t in r3 d0 d1 g8 g12 g16 g20 g30 ps a2 a32 a33 k0 k55 k2; do L="$L $t=_diag/liblfsr_w4_$t.so"; done
timeout -k 10 600 python tools/conv_ab.py base=$P/liblfsr_hip.so $L > gpurun_out/r4/c2_conv_ab.log 2>&1 || { tail -30 gpurun_out/r4/c2_conv_ab.log; exit 1; }
grep -v "amdgpu.ids\|^check" gpurun_out/r4/c2_conv_ab.log
grep "^check" gpurun_out/r4/c2_conv_ab.log | grep -v "bit-equal" | grep -v " a2 \| a32 \| a33 \| k55 \| k2 " || true
timeout -k 10 600 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu -k "conv3x3 or batch32 or residual" > gpurun_out/r4/c2_tests.log 2>&1 || { tail -30 gpurun_out/r4/c2_tests.log; exit 1; }
tail -3 gpurun_out/r4/c2_tests.log
