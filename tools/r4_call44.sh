#!/bin/bash
# round 4, call 44: LFT window attention with an XCD walking a contiguous range of (sequence, strip) pairs: tests, LFT lines of both orders on one box, FETCH_SIZE of the kernel
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "lft or window or attention" > gpurun_out/r4/c44_tests.log 2>&1 || { tail -40 gpurun_out/r4/c44_tests.log; exit 1; }
tail -2 gpurun_out/r4/c44_tests.log
for i in 1 2; do
  for tag in rr ranges; do
    if [ $tag = rr ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_win_attn_mfma_rr.so; else unset LFSR_HIP_LIB; fi
    python bench.py --workload lft --steps 6 > gpurun_out/r4/c44_lft_${tag}_$i.json 2>> gpurun_out/r4/c44_err.log
    python -c "
import json; l=json.load(open('gpurun_out/r4/c44_lft_${tag}_$i.json')); print('$tag $i LFT', round(l['value'],1), round(l['ms_per_step'],3))"
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in rr ranges; do
  if [ $tag = rr ]; then export LFSR_HIP_LIB=$R/_diag/liblfsr_win_attn_mfma_rr.so; else unset LFSR_HIP_LIB; fi
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r4/c44_pmc_$tag -o r04 --output-format csv -- python3 $R/bench.py --workload lft --steps 1 --warmup 1 > $R/gpurun_out/r4/c44_pmc_$tag.log 2>&1
  python3 - <<PY
import csv, glob
f=glob.glob('$R/gpurun_out/r4/c44_pmc_$tag/**/*counter_collection.csv', recursive=True)[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_win_attn_mfma' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
import collections
print('$tag FETCH_SIZE per launch (KB, x2 on gfx950):', round(sum(v)/max(1,len(v)/1),0) if False else None)
d=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'k_win_attn_mfma' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': d[r['Dispatch_Id']]+=float(r['Counter_Value'])
vals=sorted(d.values()); print('$tag win_attn launches', len(vals), 'read MB per launch (2 x FETCH_SIZE x 1024):', round(2*1024*sum(vals)/len(vals)/1e6,1))
PY
done
