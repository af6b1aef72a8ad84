#!/bin/bash
export LFSR_LAB=1   # (A/B selectors of the library are live only under LFSR_LAB)
# round 3, call 3: EPI branch on the three-term bf16 pipe: operator tests, accuracy sweep, whole-model tests, A/B bench in one process environment
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_bwd_ops.py tests/test_gpu_distgssr.py tests/test_gpu_lft.py -x -q -m gpu -s > gpurun_out/r3/c3_tests.log 2>&1 || { tail -40 gpurun_out/r3/c3_tests.log; exit 1; }
grep -E "three-term|passed|failed|EPI branch" gpurun_out/r3/c3_tests.log | tail -12
for sel in b3 wino b3 wino; do
  if [ $sel = wino ]; then export LFSR_EPI=wino; else unset LFSR_EPI; fi
  python bench.py --no-cpu-baseline --no-other-workloads --steps 20 > gpurun_out/r3/c3_bench_$sel.json 2>> gpurun_out/r3/c3_bench.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r3/c3_bench_$sel.json"))
print("$sel", round(j["value"],1), round(j["ms_per_step"],3), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
PY
done
unset LFSR_EPI
python -m pytest tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c3_train_tests.log 2>&1 || { tail -40 gpurun_out/r3/c3_train_tests.log; exit 1; }
tail -3 gpurun_out/r3/c3_train_tests.log
