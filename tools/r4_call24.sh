#!/bin/bash
# round 4, call 24: k_up_tail4 (matrix and VALU phases of the up-sampling tail overlapped): tail / model tests, EPIT and LFT with the previous and the new library alternating
set -e
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_epit.py tests/test_gpu_lft.py -x -q -m gpu > gpurun_out/r4/c24_tests.log 2>&1 || { tail -30 gpurun_out/r4/c24_tests.log; exit 1; }
tail -2 gpurun_out/r4/c24_tests.log
for rep in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_prev.so; else unset LFSR_HIP_LIB; fi
    for wl in epit lft; do
      python bench.py --workload $wl --steps 20 > gpurun_out/r4/c24_${wl}_${lib}_$rep.json 2>> gpurun_out/r4/c24_err.log
      python -c "
import json; j=json.load(open('gpurun_out/r4/c24_${wl}_${lib}_$rep.json')); print('$wl $lib $rep', round(j['value'],1), round(j['ms_per_step'],3))"
    done
  done
done
