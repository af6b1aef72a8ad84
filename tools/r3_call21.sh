#!/bin/bash
export LFSR_LAB=1
set -e
mkdir -p gpurun_out/r3
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export LFSR_LNLIN=0; else unset LFSR_LNLIN; fi
  python bench.py --workload epit --no-other-workloads > gpurun_out/r3/c21_epit_${v}_$i.json 2>> gpurun_out/r3/c21.err
  python bench.py --workload lft --no-other-workloads > gpurun_out/r3/c21_lft_${v}_$i.json 2>> gpurun_out/r3/c21.err
  python -c "
import json
print('$v $i', [ (json.load(open('gpurun_out/r3/c21_%s_${v}_$i.json' % w))['value']) for w in ('epit','lft')])"
done; done
