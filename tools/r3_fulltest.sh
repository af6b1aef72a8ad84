#!/bin/bash
# the driver's round-end checks, rehearsed: full GPU suite, smoke, default bench
set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/full_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3/full_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r3/full_bench.json 2> gpurun_out/r3/full_bench.err; python -c "
import json; j=json.load(open('gpurun_out/r3/full_bench.json')); print(j['value'], j['ms_per_step'], j['all_fp32_mfma']['value'], [ (round(o['value'],1), round(o['ms_per_step'],2)) for o in j['other_workloads']], j['cpu_baseline']['value'])"
