#!/bin/bash
# round 4, call 5: light stamps (none around the consumers' barrier A) of the current conv kernel, of its no-transform ablation and of the consumer-only ablation
set -e
mkdir -p gpurun_out/r4
for t in diag diag2 diag55; do
  echo "== $t" >> gpurun_out/r4/c5_stamps.log
  timeout -k 10 300 python tools/conv_stamp.py _diag/liblfsr_w4_$t.so >> gpurun_out/r4/c5_stamps.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r4/c5_stamps.log | grep -v " 0 cyc"
