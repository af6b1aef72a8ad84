#!/bin/bash
# round 3, call 8: persistent window attention; ablation timings of k_epi_b3
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py tests/test_gpu_epit.py -x -q -m gpu > gpurun_out/r3/c8_tests.log 2>&1 || { tail -40 gpurun_out/r3/c8_tests.log; exit 1; }
tail -2 gpurun_out/r3/c8_tests.log
for i in 1 2; do
python bench.py --workload lft --steps 8 > gpurun_out/r3/c8_lft_$i.json 2>> gpurun_out/r3/c8_bench.err
python bench.py --workload epit --steps 20 > gpurun_out/r3/c8_epit_$i.json 2>> gpurun_out/r3/c8_bench.err
python - <<PY
import json
for f in ("c8_lft_$i", "c8_epit_$i"):
    j=json.load(open("gpurun_out/r3/%s.json" % f)); print(f, round(j["value"],1), round(j["ms_per_step"],2))
PY
done
python tools/epi_time.py > gpurun_out/r3/c8_epi_abl.log 2>&1
for t in a1 a2 a4 a8 a16 a32 a15 a47 a63; do LFSR_HIP_LIB=$PWD/_diag/liblfsr_epi_b3_$t.so python tools/epi_time.py >> gpurun_out/r3/c8_epi_abl.log 2>&1; done
python tools/epi_time.py >> gpurun_out/r3/c8_epi_abl.log 2>&1
grep -v amdgpu.ids gpurun_out/r3/c8_epi_abl.log
