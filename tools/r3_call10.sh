#!/bin/bash
# round 3, call 10: two-stream branch backward (training), A/B in one call
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_distgssr_train.py tests/test_gpu_bwd_ops.py -x -q -m gpu > gpurun_out/r3/c10_tests.log 2>&1 || { tail -40 gpurun_out/r3/c10_tests.log; exit 1; }
tail -2 gpurun_out/r3/c10_tests.log
for i in 1 2; do
python bench.py --workload train --steps 10 > gpurun_out/r3/c10_train_$i.json 2>> gpurun_out/r3/c10_bench.err
LFSR_BWD_OVERLAP=0 python bench.py --workload train --steps 10 > gpurun_out/r3/c10_train_one_$i.json 2>> gpurun_out/r3/c10_bench.err
LFSR_BWD_OVERLAP=6 python bench.py --workload train --steps 10 > gpurun_out/r3/c10_train_both_$i.json 2>> gpurun_out/r3/c10_bench.err
python - <<PY
import json
for f in ("c10_train_$i", "c10_train_one_$i", "c10_train_both_$i"):
    j=json.load(open("gpurun_out/r3/%s.json" % f)); print(f, round(j["value"],1), round(j["ms_per_step"],2), j["loss"])
PY
done
