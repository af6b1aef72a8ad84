#!/bin/bash
set -e
mkdir -p gpurun_out/r3
for t in product v1 v2 v4 v7; do
  if [ $t = product ]; then unset LFSR_HIP_LIB; else export LFSR_HIP_LIB=$PWD/_diag/liblfsr_epi_b3_$t.so; fi
  echo "== $t" >> gpurun_out/r3/c9b.log
  python -m pytest tests/test_gpu_b3_accuracy.py -q -m gpu -k epi_branch -s 2>&1 | grep -E "EPI branch|passed|failed" >> gpurun_out/r3/c9b.log || true
done
cat gpurun_out/r3/c9b.log
