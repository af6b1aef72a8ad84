#!/bin/bash
# round 3, call 2: new tests + gradient-parity decomposition + default bench with other_workloads
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_bench_ranks.py "tests/test_gpu_distgssr_train.py::test_grads_full_geometry_vs_torch_port_autograd" -x -q -m gpu -s > gpurun_out/r3/c2_tests.log 2>&1 || { tail -40 gpurun_out/r3/c2_tests.log; exit 1; }
tail -12 gpurun_out/r3/c2_tests.log
python bench.py > gpurun_out/r3/c2_bench.json 2> gpurun_out/r3/c2_bench.err || { tail -20 gpurun_out/r3/c2_bench.err; exit 1; }
python tools/grad_parity.py gpurun_out/r3/grad_parity.json > gpurun_out/r3/grad_parity.log 2>&1 || { tail -20 gpurun_out/r3/grad_parity.log; exit 1; }
grep -v "^CPU" gpurun_out/r3/grad_parity.log | tail -30
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3/c2_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"])
for o in j["other_workloads"]:
    d=o["dominant_kernel"]
    print(o["config"][:40], round(o["value"],1), round(o["ms_per_step"],2), d["operator"], d["tags"], round(d["avg_launch_us"],1), d.get("frac"), (o.get("all_fp32_mfma") or {}).get("value"))
PY
