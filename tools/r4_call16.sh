#!/bin/bash
# round 4, call 16: a3 with one input position per thread (coalesced 16-B stores), 32-bit index decode in k_initconv / k_head: index tests, class lines of the bench
set -e
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_index_ops.py tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r4/c16_tests.log 2>&1 || { tail -30 gpurun_out/r4/c16_tests.log; exit 1; }
tail -2 gpurun_out/r4/c16_tests.log
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r4/c16_bench.json 2> gpurun_out/r4/c16_err.log || { tail gpurun_out/r4/c16_err.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4/c16_bench.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3))
for c in j["roofline_classes"]: print(c["class"][:48].ljust(48), round(c["us"],1), "us", c.get("unit"), round(c.get("achieved",0),2), "frac", round(c.get("frac",0),3), "of copy", round(c.get("frac_of_copy",0),3))
PY
