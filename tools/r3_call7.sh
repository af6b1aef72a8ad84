#!/bin/bash
# round 3, call 7: k_epi_b3 third form (per-view barriers, DPP shift, t registers as stage-2 operand)
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_distgssr.py tests/test_gpu_bwd_ops.py -x -q -m gpu > gpurun_out/r3/c7_tests.log 2>&1 || { tail -40 gpurun_out/r3/c7_tests.log; exit 1; }
tail -2 gpurun_out/r3/c7_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r3/c7_bench_$i.json 2>> gpurun_out/r3/c7_bench.err
python - <<PY
import json
j=json.load(open("gpurun_out/r3/c7_bench_$i.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
PY
done
python -m pytest tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c7_train_tests.log 2>&1 || { tail -40 gpurun_out/r3/c7_train_tests.log; exit 1; }
tail -2 gpurun_out/r3/c7_train_tests.log
python bench.py --workload train --steps 10 > gpurun_out/r3/c7_train.json 2>> gpurun_out/r3/c7_bench.err; python -c "
import json; j=json.load(open('gpurun_out/r3/c7_train.json')); print('train', j['value'], j['ms_per_step'])"
