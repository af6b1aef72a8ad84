#!/bin/bash
# round 4, call 11: training step on ONE stream (LFSR_BWD_OVERLAP=0): isolated kernel times of the branch gradients
set -e
mkdir -p gpurun_out/r4
export LFSR_LAB=1 LFSR_BWD_OVERLAP=0
python bench.py --workload train --steps 10 > gpurun_out/r4/c11_train_1stream.json 2>> gpurun_out/r4/c11_err.log; python -c "
import json; j=json.load(open('gpurun_out/r4/c11_train_1stream.json')); print('train one stream', round(j['ms_per_step'],3), 'ms')"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4/c11_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/c11_trace -o train --output-format csv -- python3 bench.py --workload train --steps 8 --warmup 2 > /dev/null 2>> gpurun_out/r4/c11_err.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4/c11_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step ms", tot / 10 / 1e6)
for r in rows[:24]:
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    print(f"{r['Name'][:80]:80s} calls {n:5d} avg {t/n/1e3:8.1f} us  {100*t/tot:5.1f}%")
PY
