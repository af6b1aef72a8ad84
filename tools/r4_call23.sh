#!/bin/bash
# round 4, call 23: k_ang_fused with the second row of views requested before the weight staging's barrier: class line of the bench, old / new alternating; ang tests
set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu -k "ang" > gpurun_out/r4/c23_tests.log 2>&1 || { tail -20 gpurun_out/r4/c23_tests.log; exit 1; }
tail -1 gpurun_out/r4/c23_tests.log
for rep in 1 2 3; do
  for lib in pre1 new; do
    if [ $lib = pre1 ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_ang_fused_pre1.so; else unset LFSR_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-other-workloads --no-split-check --steps 10 > gpurun_out/r4/c23_$lib.json 2>> gpurun_out/r4/c23_err.log
    python -c "
import json; j=json.load(open('gpurun_out/r4/c23_$lib.json')); print('$lib', round(j['value'],1), 'ang ms', round(j['kernel_ms_per_step']['angconv'],4))"
  done
done
