#!/bin/bash
# round 4, call 37: the up-sampling tail at several batch sizes (is B = 8 slower per patch than 64?)
set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/uptail_time.py > gpurun_out/r4/c37_uptail.log 2>&1 || { tail -20 gpurun_out/r4/c37_uptail.log; exit 1; }
cat gpurun_out/r4/c37_uptail.log
