#!/bin/bash
# round 4, call 27: training step with the clip on the flat bucket and the fused AdamW: training tests, the step
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_distgssr_train.py tests/test_gpu_bench_ranks.py -x -q -m gpu > gpurun_out/r4/c27_tests.log 2>&1 || { tail -40 gpurun_out/r4/c27_tests.log; exit 1; }
tail -2 gpurun_out/r4/c27_tests.log
for i in 1 2 3; do python bench.py --workload train --steps 20 > gpurun_out/r4/c27_train_$i.json 2>> gpurun_out/r4/c27_err.log; python -c "
import json; j=json.load(open('gpurun_out/r4/c27_train_$i.json')); print('train', round(j['ms_per_step'],3), 'ms', round(j['value'],1), 'loss', j['loss'])"; done
