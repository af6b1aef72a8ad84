#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_epit.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --workload epit --no-other-workloads > gpurun_out/r3/c27_epit_new_$i.json 2>> gpurun_out/r3/c27.err
  LFSR_HIP_LIB=$PWD/_diag/liblfsr_attn_mfma_old.so python bench.py --workload epit --no-other-workloads > gpurun_out/r3/c27_epit_old_$i.json 2>> gpurun_out/r3/c27.err
  python -c "
import json
print('$i', [ (json.load(open('gpurun_out/r3/c27_epit_%s_$i.json' % w))['value']) for w in ('new','old')])"
done
