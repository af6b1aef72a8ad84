#!/bin/bash
# round 4, call 25: kernel durations of the up-sampling tail, fourth against third form (rocprofv3 kernel trace of the EPIT and LFT bench lines)
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 4; do
  for wl in epit lft; do
    rm -rf gpurun_out/r4/c25_$v$wl
    if [ $v = 3 ]; then export LFSR_LAB=1 LFSR_UPTAIL=3; else unset LFSR_UPTAIL; fi
    rocprofv3 --kernel-trace --stats -d gpurun_out/r4/c25_$v$wl -o t --output-format csv -- python3 bench.py --workload $wl --steps 5 --warmup 2 > /dev/null 2>> gpurun_out/r4/c25_err.log
    python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/r4/c25_$v$wl/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "up_tail" in r["Name"]: print("form $v $wl", r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
  done
done
timeout -k 10 600 python -m pytest tests/test_gpu_epit.py tests/test_gpu_lft.py -x -q -m gpu > gpurun_out/r4/c25_tests.log 2>&1 || { tail -30 gpurun_out/r4/c25_tests.log; exit 1; }
tail -1 gpurun_out/r4/c25_tests.log
