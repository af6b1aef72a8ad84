#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_bwd_ops.py tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c19_tests.log 2>&1 || { tail -40 gpurun_out/r3/c19_tests.log; exit 1; }
tail -2 gpurun_out/r3/c19_tests.log
for i in 1 2; do
python bench.py --workload train --steps 10 > gpurun_out/r3/c19_train_$i.json 2>> gpurun_out/r3/c19.err
LFSR_DGRAD_PW=f32 python bench.py --workload train --steps 10 > gpurun_out/r3/c19_train_f32_$i.json 2>> gpurun_out/r3/c19.err
python -c "
import json
for f in ('c19_train_$i','c19_train_f32_$i'):
    j=json.load(open('gpurun_out/r3/%s.json' % f)); print(f, j['value'], j['ms_per_step'], j['loss'])"; done
