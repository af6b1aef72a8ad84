#!/bin/bash
# round 4, call 30: half tiles for the conv's last partial round: conv / model tests, A/B against HEAD in one process
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r4/c30_tests.log 2>&1 || { tail -60 gpurun_out/r4/c30_tests.log; exit 1; }
tail -2 gpurun_out/r4/c30_tests.log
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=6 AB_GEOMS="800:n,800:y,200:n,25:n" timeout -k 10 300 python tools/conv_ab.py head=_diag/liblfsr_w4_head.so new=$P/liblfsr_hip.so > gpurun_out/r4/c30_ab.log 2>&1 || { tail -20 gpurun_out/r4/c30_ab.log; exit 1; }
cat gpurun_out/r4/c30_ab.log
