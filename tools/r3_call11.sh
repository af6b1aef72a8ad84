#!/bin/bash
# round 3, call 11: conv drain without the activation's VALU when slope == 1; init_conv with its weights in registers
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_distgssr.py tests/test_gpu_bwd_ops.py tests/test_gpu_internet.py -x -q -m gpu > gpurun_out/r3/c11_tests.log 2>&1 || { tail -40 gpurun_out/r3/c11_tests.log; exit 1; }
tail -2 gpurun_out/r3/c11_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r3/c11_bench_$i.json 2>> gpurun_out/r3/c11_bench.err
python - <<PY
import json
j=json.load(open("gpurun_out/r3/c11_bench_$i.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()}, round(j["roofline"]["avg_launch_us"],1))
PY
done
python bench.py --workload train --steps 10 > gpurun_out/r3/c11_train.json 2>> gpurun_out/r3/c11_bench.err; python -c "
import json; j=json.load(open('gpurun_out/r3/c11_train.json')); print('train', j['value'], j['ms_per_step'])"
