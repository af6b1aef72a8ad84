#!/bin/bash
# round 4, call 42: cache policy of the conv's output stores inside the whole forward (the next conv reads what this one wrote): nt (default) against plain / sc0 / sc0+nt
set -e
mkdir -p gpurun_out/r4
for i in 1 2; do
  for tag in nt st0 st1 st3; do
    if [ $tag = nt ]; then unset LFSR_HIP_LIB; else export LFSR_HIP_LIB=$PWD/_diag/liblfsr_w4_$tag.so; fi
    python bench.py --steps 20 --no-cpu-baseline --no-other-workloads --no-split-check > gpurun_out/r4/c42_bench_${tag}_$i.json 2>> gpurun_out/r4/c42_err.log
    python -c "
import json; j=json.load(open('gpurun_out/r4/c42_bench_${tag}_$i.json')); print('$tag $i headline', round(j['value'],1), round(j['ms_per_step'],3), 'conv', round(j['roofline']['avg_launch_us'],1))"
  done
done
