#!/bin/bash
# round 4, call 13: conv drain with wave-uniform (SGPR) plane offsets on aligned geometries: A/B against the previous build, conv tests
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=8 timeout -k 10 300 python tools/conv_ab.py prev=_diag/liblfsr_attn_old.so new=$P/liblfsr_hip.so r3=_diag/liblfsr_w4_r3.so > gpurun_out/r4/c13_conv_ab.log 2>&1 || { tail -20 gpurun_out/r4/c13_conv_ab.log; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r4/c13_conv_ab.log | grep -v "^check.*bit-equal" 
timeout -k 10 600 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r4/c13_tests.log 2>&1 || { tail -30 gpurun_out/r4/c13_tests.log; exit 1; }
tail -2 gpurun_out/r4/c13_tests.log
