#!/bin/bash
export LFSR_LAB=1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_lft.py -x -q -m gpu > gpurun_out/r3/c25_tests.log 2>&1 || { tail -40 gpurun_out/r3/c25_tests.log; exit 1; }
tail -2 gpurun_out/r3/c25_tests.log
O=$R/gpurun_out/r3/c25; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/new -o r03 --output-format csv -- python3 $R/bench.py --workload lft --steps 3 --warmup 1 --no-other-workloads > $O/new.log 2>&1 || exit 1
for v in new; do f=$(find $O/$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -14 $f | cut -c1-150; done
tail -1 $O/new.log | cut -c1-120
