#!/bin/bash
export LFSR_LAB=1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r3/c24; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/new -o r03 --output-format csv -- python3 $R/bench.py --workload lft --steps 3 --warmup 1 --no-other-workloads > $O/new.log 2>&1 || exit 1
export LFSR_FFN=chunks
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/old -o r03 --output-format csv -- python3 $R/bench.py --workload lft --steps 3 --warmup 1 --no-other-workloads > $O/old.log 2>&1 || exit 1
for v in new old; do f=$(find $O/$v -name "*kernel_stats.csv" | head -1); echo "== $v"; grep -i "ffn" $f | cut -c1-140; done
