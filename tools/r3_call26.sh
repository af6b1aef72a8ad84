#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py tests/test_gpu_epit.py tests/test_gpu_dispatch.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export LFSR_ATTN_ANG=lds; else unset LFSR_ATTN_ANG; fi
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c26_lft_${v}_$i.json 2>> gpurun_out/r3/c26.err
  python -c "
import json
print('$v $i', json.load(open('gpurun_out/r3/c26_lft_${v}_$i.json'))['value'])"
done; done
