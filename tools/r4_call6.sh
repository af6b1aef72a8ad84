#!/bin/bash
# round 4, call 6: deeper V fragment ring (LDS latency tolerance of the consumers), producer-only barrier A on the new step order, barrier-A position
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
L=""
for t in v6u9 v9u9 u9 ps g12 g30 g34; do L="$L $t=_diag/liblfsr_w4_$t.so"; done
AB_ROUNDS=8 timeout -k 10 600 python tools/conv_ab.py base=$P/liblfsr_hip.so $L > gpurun_out/r4/c6_conv_ab.log 2>&1 || { tail -30 gpurun_out/r4/c6_conv_ab.log; exit 1; }
grep -v "amdgpu.ids\|^check" gpurun_out/r4/c6_conv_ab.log
grep "^check" gpurun_out/r4/c6_conv_ab.log | grep -v "bit-equal" || true
