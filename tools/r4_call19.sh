#!/bin/bash
# round 4, call 19: patches/s of the DistgSSR forward by batch size: do sub-batches whose tensors stay in the 256-MiB memory-side cache beat B = 32?
mkdir -p gpurun_out/r4
for B in 32 9 10 11 12 16 20 23 32; do
  python bench.py --batch $B --steps 20 --no-cpu-baseline --no-other-workloads --no-split-check > gpurun_out/r4/c19_b$B.json 2>> gpurun_out/r4/c19_err.log
  python -c "
import json; j=json.load(open('gpurun_out/r4/c19_b$B.json')); k=j['kernel_ms_per_step']; print('B $B', round(j['value'],1), 'patches/s', round(j['ms_per_step'],3), 'ms  conv us', round(j['roofline']['avg_launch_us'],1), {a: round(b,3) for a,b in k.items()})"
done
