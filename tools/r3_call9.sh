#!/bin/bash
export LFSR_LAB=1   # (A/B selectors of the library are live only under LFSR_LAB)
# round 3, call 9: k_epi_b3 with phase-shifted split bursts and a leaner tail
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_b3_accuracy.py tests/test_gpu_distgssr.py tests/test_gpu_bwd_ops.py -x -q -m gpu > gpurun_out/r3/c9_tests.log 2>&1 || { tail -40 gpurun_out/r3/c9_tests.log; exit 1; }
tail -2 gpurun_out/r3/c9_tests.log
python tools/epi_time.py > gpurun_out/r3/c9_epi_abl.log 2>&1
for t in a1 a8 a16 a63 v4; do LFSR_HIP_LIB=$PWD/_diag/liblfsr_epi_b3_$t.so python tools/epi_time.py >> gpurun_out/r3/c9_epi_abl.log 2>&1; done
LFSR_EPI=wino python tools/epi_time.py >> gpurun_out/r3/c9_epi_abl.log 2>&1
python tools/epi_time.py >> gpurun_out/r3/c9_epi_abl.log 2>&1
grep -v amdgpu.ids gpurun_out/r3/c9_epi_abl.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r3/c9_bench_$i.json 2>> gpurun_out/r3/c9_bench.err
python - <<PY
import json
j=json.load(open("gpurun_out/r3/c9_bench_$i.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
PY
done
