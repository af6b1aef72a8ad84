#!/bin/bash
# round 4, call 18: up_tail3's halo columns as 16-B LDS reads: EPIT / LFT lines with the previous and the new library, alternating; tail tests
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
timeout -k 10 600 python -m pytest tests/test_gpu_epit.py tests/test_gpu_lft.py -x -q -m gpu -k "tail or forward or model or full or epit or lft" > gpurun_out/r4/c18_tests.log 2>&1 || { tail -30 gpurun_out/r4/c18_tests.log; exit 1; }
tail -2 gpurun_out/r4/c18_tests.log
for rep in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_prev.so; else unset LFSR_HIP_LIB; fi
    for wl in epit lft; do
      python bench.py --workload $wl --steps 20 > gpurun_out/r4/c18_${wl}_${lib}_$rep.json 2>> gpurun_out/r4/c18_err.log
      python -c "
import json; j=json.load(open('gpurun_out/r4/c18_${wl}_${lib}_$rep.json')); print('$wl $lib $rep', round(j['value'],1), round(j['ms_per_step'],3))"
    done
  done
done
