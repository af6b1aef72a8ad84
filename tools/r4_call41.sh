#!/bin/bash
# round 4, call 41: is the conv slower inside a forward because its U fragments are cold in L2?  one weight pack reused against 16 packs used in turn
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
for nw in 1 16 1 16; do
  echo "AB_NW=$nw"
  AB_NW=$nw AB_ROUNDS=4 AB_GEOMS="800:n,800:y,200:n" timeout -k 10 300 python tools/conv_ab.py new=$P/liblfsr_hip.so 2>&1 | grep -v "^check\|amdgpu.ids"
done > gpurun_out/r4/c41_cold_u.log 2>&1
cat gpurun_out/r4/c41_cold_u.log
