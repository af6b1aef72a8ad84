#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c13_tests.log 2>&1 || { tail -40 gpurun_out/r3/c13_tests.log; exit 1; }
tail -2 gpurun_out/r3/c13_tests.log
LFSR_BWD_OVERLAP=10 python -m pytest tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c13_tests10.log 2>&1 || { tail -40 gpurun_out/r3/c13_tests10.log; exit 1; }
tail -2 gpurun_out/r3/c13_tests10.log
for i in 1 2; do
python bench.py --workload train --steps 10 > gpurun_out/r3/c13_train_$i.json 2>> gpurun_out/r3/c13.err
LFSR_BWD_OVERLAP=10 python bench.py --workload train --steps 10 > gpurun_out/r3/c13_train_red_$i.json 2>> gpurun_out/r3/c13.err
python -c "
import json
for f in ('c13_train_$i','c13_train_red_$i'):
    j=json.load(open('gpurun_out/r3/%s.json' % f)); print(f, j['value'], j['ms_per_step'], j['loss'])"; done
