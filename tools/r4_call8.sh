#!/bin/bash
# round 4, call 8: full GPU suite after the cleanup / bench changes, then the default bench line
set -e
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/c8_fulltest.log 2>&1 || { tail -40 gpurun_out/r4/c8_fulltest.log; exit 1; }
tail -3 gpurun_out/r4/c8_fulltest.log
python bench.py > gpurun_out/r4/c8_bench.json 2> gpurun_out/r4/c8_bench.err || { tail -20 gpurun_out/r4/c8_bench.err; exit 1; }
python - <<PY
import json
j=json.load(open("gpurun_out/r4/c8_bench.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), "allf32", round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
print("roofline frac", round(j["roofline"]["frac"],3), "conv us", round(j["roofline"]["avg_launch_us"],1))
for w in j.get("other_workloads", []): print(w["config"][:60], round(w["value"],1), round(w["ms_per_step"],3), "f32:", round(w.get("all_fp32_mfma",{}).get("value",0),1))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
