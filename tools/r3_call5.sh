#!/bin/bash
export LFSR_LAB=1   # (A/B selectors of the library are live only under LFSR_LAB)
# round 3, call 5: software-pipelined split in k_epi_b3; batched staging in k_win_attn_mfma
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py tests/test_gpu_b3_accuracy.py tests/test_gpu_distgssr.py -x -q -m gpu > gpurun_out/r3/c5_tests.log 2>&1 || { tail -40 gpurun_out/r3/c5_tests.log; exit 1; }
tail -2 gpurun_out/r3/c5_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r3/c5_bench_$i.json 2>> gpurun_out/r3/c5_bench.err
python - <<PY
import json
j=json.load(open("gpurun_out/r3/c5_bench_$i.json"))
print("headline", round(j["value"],1), round(j["ms_per_step"],3), round(j["all_fp32_mfma"]["value"],1), {k: round(v,3) for k,v in j["kernel_ms_per_step"].items()})
PY
python bench.py --workload lft --steps 8 > gpurun_out/r3/c5_lft_$i.json 2>> gpurun_out/r3/c5_bench.err
LFSR_ATTN=valu python bench.py --workload lft --steps 8 > gpurun_out/r3/c5_lft_valu_$i.json 2>> gpurun_out/r3/c5_bench.err
python - <<PY
import json
for f in ("c5_lft_$i", "c5_lft_valu_$i"):
    j=json.load(open("gpurun_out/r3/%s.json" % f)); print(f, round(j["value"],1), round(j["ms_per_step"],2))
PY
done
