#!/bin/bash
# round 4, call 31: half tiles paired on an XCD: conv tests, A/B against HEAD
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu -k "conv3x3 or batch32 or packed" > gpurun_out/r4/c31_tests.log 2>&1 || { tail -60 gpurun_out/r4/c31_tests.log; exit 1; }
tail -2 gpurun_out/r4/c31_tests.log
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=6 AB_GEOMS="800:n,800:y,200:n,25:n" timeout -k 10 300 python tools/conv_ab.py head=_diag/liblfsr_w4_head.so new=$P/liblfsr_hip.so > gpurun_out/r4/c31_ab.log 2>&1 || { tail -20 gpurun_out/r4/c31_ab.log; exit 1; }
grep -v "^check" gpurun_out/r4/c31_ab.log
