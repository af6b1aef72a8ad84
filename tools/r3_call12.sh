#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
for i in 1 2; do
python bench.py --workload lft --steps 8 > gpurun_out/r3/c12_lft_$i.json 2>> gpurun_out/r3/c12.err
LFSR_ATTN_ANG=loop python bench.py --workload lft --steps 8 > gpurun_out/r3/c12_lft_loop_$i.json 2>> gpurun_out/r3/c12.err
python -c "
import json
for f in ('c12_lft_$i','c12_lft_loop_$i'):
    j=json.load(open('gpurun_out/r3/%s.json' % f)); print(f, j['value'], j['ms_per_step'])"; done
