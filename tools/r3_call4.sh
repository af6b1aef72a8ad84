#!/bin/bash
# round 3, call 4: training tests, fuse.0 accuracy, default bench (headline + all_fp32_mfma + other_workloads + cpu baseline)
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_distgssr_train.py tests/test_gpu_b3_accuracy.py tests/test_gpu_epit.py -x -q -m gpu > gpurun_out/r3/c4_tests.log 2>&1 || { tail -40 gpurun_out/r3/c4_tests.log; exit 1; }
tail -3 gpurun_out/r3/c4_tests.log
python bench.py > gpurun_out/r3/c4_bench.json 2> gpurun_out/r3/c4_bench.err || { tail -20 gpurun_out/r3/c4_bench.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3/c4_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["all_fp32_mfma"]["value"])
for o in j["other_workloads"]:
    d=o["dominant_kernel"]
    print(o["config"][:40], round(o["value"],1), round(o["ms_per_step"],2), d["operator"], d["tags"], round(d["avg_launch_us"],1), d.get("frac"), (o.get("all_fp32_mfma") or {}).get("value"))
    print("    ", d["by_operator_ms_per_step"])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
