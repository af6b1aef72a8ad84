#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_lft.py tests/test_gpu_epit.py -x -q -m gpu > gpurun_out/r3/c23_tests.log 2>&1 || { tail -40 gpurun_out/r3/c23_tests.log; exit 1; }
tail -2 gpurun_out/r3/c23_tests.log
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export LFSR_FFN=chunks; else unset LFSR_FFN; fi
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c23_lft_${v}_$i.json 2>> gpurun_out/r3/c23.err
  python -c "
import json
print('$v $i', json.load(open('gpurun_out/r3/c23_lft_${v}_$i.json'))['value'])"
done; done
