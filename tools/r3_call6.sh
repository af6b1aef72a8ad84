#!/bin/bash
# round 3, call 6: backward overlap (wgrad on a side stream), attention staging fixes, PMC counters of the new kernels
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py tests/test_gpu_epit.py tests/test_gpu_distgssr_train.py -x -q -m gpu > gpurun_out/r3/c6_tests.log 2>&1 || { tail -40 gpurun_out/r3/c6_tests.log; exit 1; }
tail -2 gpurun_out/r3/c6_tests.log
for i in 1 2; do
python bench.py --workload train --steps 10 > gpurun_out/r3/c6_train_$i.json 2>> gpurun_out/r3/c6_bench.err
LFSR_BWD_OVERLAP=0 python bench.py --workload train --steps 10 > gpurun_out/r3/c6_train_noov_$i.json 2>> gpurun_out/r3/c6_bench.err
python bench.py --workload lft --steps 8 > gpurun_out/r3/c6_lft_$i.json 2>> gpurun_out/r3/c6_bench.err
python bench.py --workload epit --steps 20 > gpurun_out/r3/c6_epit_$i.json 2>> gpurun_out/r3/c6_bench.err
python - <<PY
import json
for f in ("c6_train_$i", "c6_train_noov_$i", "c6_lft_$i", "c6_epit_$i"):
    j=json.load(open("gpurun_out/r3/%s.json" % f)); print(f, round(j["value"],1), round(j["ms_per_step"],2))
PY
done
bash tools/r3_pmc.sh > gpurun_out/r3/c6_pmc.log 2>&1 || { tail -30 gpurun_out/r3/c6_pmc.log; exit 1; }
tail -120 gpurun_out/r3/c6_pmc.log
