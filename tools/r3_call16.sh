#!/bin/bash
set -e
mkdir -p gpurun_out/r3
python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
LFSR_HIP_LIB=$PWD/_diag/liblfsr_nodot2.so python tools/lin_time.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_epit.py tests/test_gpu_lft.py tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r3/c16_tests.log 2>&1 || { tail -40 gpurun_out/r3/c16_tests.log; exit 1; }
tail -2 gpurun_out/r3/c16_tests.log
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_nodot2.so; else unset LFSR_HIP_LIB; fi
  python bench.py --workload epit --no-other-workloads > gpurun_out/r3/c16_epit_${v}_$i.json 2>> gpurun_out/r3/c16.err
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c16_lft_${v}_$i.json 2>> gpurun_out/r3/c16.err
  python -c "
import json
print('$v $i', [ (json.load(open('gpurun_out/r3/c16_%s_${v}_$i.json' % w))['value']) for w in ('epit','lft')])"
done; done
