#!/bin/bash
# round 4, call 4: is the producers' memory traffic paid in HBM bytes or in instructions?  stores / loads redirected to a 64-KB window (cache hits), clock probes
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
L=""
for t in k0 k256 k512 k1792 k33 k2 k1794 k55; do L="$L $t=_diag/liblfsr_w4_$t.so"; done
AB_ROUNDS=6 timeout -k 10 600 python tools/conv_ab.py base=$P/liblfsr_hip.so $L > gpurun_out/r4/c4_conv_ab.log 2>&1 || { tail -30 gpurun_out/r4/c4_conv_ab.log; exit 1; }
grep -v "amdgpu.ids\|^check" gpurun_out/r4/c4_conv_ab.log
