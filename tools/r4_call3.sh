#!/bin/bash
# round 4, call 3: exchange barrier doubling as the last chunk barrier, consumers join barrier A without draining LDS; tests + headline
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
L=""
for t in r3 c2 xb anw k0; do L="$L $t=_diag/liblfsr_w4_$t.so"; done
AB_ROUNDS=8 timeout -k 10 600 python tools/conv_ab.py base=$P/liblfsr_hip.so $L > gpurun_out/r4/c3_conv_ab.log 2>&1 || { tail -30 gpurun_out/r4/c3_conv_ab.log; exit 1; }
grep -v "amdgpu.ids\|^check" gpurun_out/r4/c3_conv_ab.log
grep "^check" gpurun_out/r4/c3_conv_ab.log | grep -v "bit-equal" || true
timeout -k 10 900 python -m pytest tests/test_gpu_distgssr.py tests/test_gpu_epit.py -x -q -m gpu > gpurun_out/r4/c3_tests.log 2>&1 || { tail -30 gpurun_out/r4/c3_tests.log; exit 1; }
tail -3 gpurun_out/r4/c3_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r4/c3_bench.json 2> gpurun_out/r4/c3_bench.err || { tail -20 gpurun_out/r4/c3_bench.err; exit 1; }
python - <<PY
import json
j=json.load(open("gpurun_out/r4/c3_bench.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), "allf32", round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
print("roofline", j["roofline"])
for w in j.get("other_workloads", []): print(w.get("workload"), w.get("value"), w.get("ms_per_step"))
PY
