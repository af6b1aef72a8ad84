#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py tests/test_gpu_epit.py -x -q -m gpu > gpurun_out/r3/c22_tests.log 2>&1 || { tail -40 gpurun_out/r3/c22_tests.log; exit 1; }
tail -2 gpurun_out/r3/c22_tests.log
for i in 1 2; do
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c22_lft_new_$i.json 2>> gpurun_out/r3/c22.err
  LFSR_HIP_LIB=$PWD/_diag/liblfsr_win_attn_mfma_old.so python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c22_lft_old_$i.json 2>> gpurun_out/r3/c22.err
  python -c "
import json
print('$i', [ (json.load(open('gpurun_out/r3/c22_lft_%s_$i.json' % w))['value']) for w in ('new','old')])"
done
