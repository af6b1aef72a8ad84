#!/bin/bash
# round 4, call 33: bench lines of HEAD's conv kernel and of the half-tile one on the same box, alternating (separate processes: same device)
set -e
mkdir -p gpurun_out/r4
for i in 1 2; do
  for tag in head new; do
    if [ $tag = head ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_w4_head.so; else unset LFSR_HIP_LIB; fi
    python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r4/c33_bench_${tag}_$i.json 2>> gpurun_out/r4/c33_err.log
    python bench.py --workload train --steps 20 > gpurun_out/r4/c33_train_${tag}_$i.json 2>> gpurun_out/r4/c33_err.log
    python - <<PY
import json
j=json.load(open('gpurun_out/r4/c33_bench_${tag}_$i.json')); t=json.load(open('gpurun_out/r4/c33_train_${tag}_$i.json'))
print('$tag $i headline', round(j['value'],1), round(j['ms_per_step'],3), 'conv', round(j['roofline']['avg_launch_us'],1), 'f32', round(j['all_fp32_mfma']['value'],1), '|', ' '.join(str(round(o['value'],1)) for o in j['other_workloads']), '| train', round(t['ms_per_step'],3))
PY
  done
done
