#!/bin/bash
# round 4, call 10: AngConv.0 data gradient with register weights and four pixels in flight: gradient tests, training step, its kernel time
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_bwd_ops.py tests/test_gpu_distgssr_train.py tests/test_gpu_limits.py -x -q -m gpu > gpurun_out/r4/c10_tests.log 2>&1 || { tail -40 gpurun_out/r4/c10_tests.log; exit 1; }
tail -3 gpurun_out/r4/c10_tests.log
for i in 1 2; do python bench.py --workload train --steps 10 > gpurun_out/r4/c10_train_$i.json 2>> gpurun_out/r4/c10_err.log; python -c "
import json; j=json.load(open('gpurun_out/r4/c10_train_$i.json')); print('train', round(j['ms_per_step'],3), 'ms', round(j['value'],1))"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4/c10_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/c10_trace -o train --output-format csv -- python3 bench.py --workload train --steps 8 --warmup 2 > /dev/null 2>> gpurun_out/r4/c10_err.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4/c10_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    print(f"{r['Name'][:80]:80s} calls {n:5d} avg {t/n/1e3:8.1f} us  {100*t/tot:5.1f}%")
PY
