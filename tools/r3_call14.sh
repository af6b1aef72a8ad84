#!/bin/bash
# A/B of the split's residual form: product (v_dot2c_f32_bf16) vs _diag/liblfsr_nodot2.so (and / sub), same box, alternating
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_epit.py tests/test_gpu_lft.py tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r3/c14_tests.log 2>&1 || { tail -40 gpurun_out/r3/c14_tests.log; exit 1; }
tail -2 gpurun_out/r3/c14_tests.log
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_nodot2.so; else unset LFSR_HIP_LIB; fi
  python bench.py --no-other-workloads > gpurun_out/r3/c14_h_${v}_$i.json 2>> gpurun_out/r3/c14.err
  python bench.py --workload epit --no-other-workloads > gpurun_out/r3/c14_epit_${v}_$i.json 2>> gpurun_out/r3/c14.err
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c14_lft_${v}_$i.json 2>> gpurun_out/r3/c14.err
  python -c "
import json
print('$v $i', [ (json.load(open('gpurun_out/r3/c14_%s_${v}_$i.json' % w))['value']) for w in ('h','epit','lft')])"
done; done
