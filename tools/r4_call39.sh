#!/bin/bash
# round 4, call 39: fused feed-forward with the two waves of a SIMD half a chunk apart (barrier at different points of the phase cycle): tests, EPIT / LFT lines of both forms on one box
set -e
mkdir -p gpurun_out/r4
for i in 1 2; do
  for tag in instep stag2; do
    export LFSR_HIP_LIB=$PWD/_diag/liblfsr_ffn_b3_$tag.so
    python bench.py --workload epit --steps 20 > gpurun_out/r4/c40_epit_${tag}_$i.json 2>> gpurun_out/r4/c40_err.log
    python bench.py --workload lft --steps 6 > gpurun_out/r4/c40_lft_${tag}_$i.json 2>> gpurun_out/r4/c40_err.log
    python - <<PY
import json
e=json.load(open('gpurun_out/r4/c40_epit_${tag}_$i.json')); l=json.load(open('gpurun_out/r4/c40_lft_${tag}_$i.json'))
print('$tag $i EPIT', round(e['value'],1), round(e['ms_per_step'],3), '| LFT', round(l['value'],1), round(l['ms_per_step'],3))
PY
  done
done
