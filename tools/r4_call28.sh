#!/bin/bash
# round 4, call 28: packed row pass of the conv's input transform (operand selectors): conv tests, A/B against the scalar row pass in one process
set -e
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_distgssr.py -x -q -m gpu -k "conv3x3 or packed or batch32" > gpurun_out/r4/c28_tests.log 2>&1 || { tail -40 gpurun_out/r4/c28_tests.log; exit 1; }
tail -2 gpurun_out/r4/c28_tests.log
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=6 timeout -k 10 300 python tools/conv_ab.py old=_diag/liblfsr_w4_old.so rowpk=_diag/liblfsr_w4_rowpk.so new=$P/liblfsr_hip.so > gpurun_out/r4/c28_ab.log 2>&1 || { tail -20 gpurun_out/r4/c28_ab.log; exit 1; }
cat gpurun_out/r4/c28_ab.log
python bench.py --steps 20 > gpurun_out/r4/c28_bench.json 2>> gpurun_out/r4/c28_err.log
python -c "
import json; j=json.load(open('gpurun_out/r4/c28_bench.json')); print('headline', j['value'], j['ms_per_step'], j['roofline'])"
