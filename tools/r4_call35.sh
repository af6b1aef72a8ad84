#!/bin/bash
# round 4, call 35: group skip fused into the first block's SpaConv.0 data gradient; the half-tile invariance test; whole suite; same-box training step
set -e
mkdir -p gpurun_out/r4
tools/r4_fulltest.sh c35_fulltest
for i in 1 2; do
  python bench.py --workload train --steps 20 > gpurun_out/r4/c35_train_$i.json 2>> gpurun_out/r4/c35_err.log
  python -c "
import json; j=json.load(open('gpurun_out/r4/c35_train_$i.json')); print('train', round(j['ms_per_step'],3), 'ms', 'loss', j['loss'])"
done
