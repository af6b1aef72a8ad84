#!/bin/bash
# round 4, call 36: row-GEMM with the next tile's rows requested in front of the split: operator tests, bench lines of both forms on one box (alternating processes)
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "pointwise or fuse0 or linear or b3 or epit or lft or distgssr_full or batch32" > gpurun_out/r4/c36_tests.log 2>&1 || { tail -40 gpurun_out/r4/c36_tests.log; exit 1; }
tail -2 gpurun_out/r4/c36_tests.log
for i in 1 2; do
  for tag in late early; do
    if [ $tag = late ]; then export LFSR_HIP_LIB=$PWD/_diag/liblfsr_rowgemm_b3_late.so; else unset LFSR_HIP_LIB; fi
    python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r4/c36_bench_${tag}_$i.json 2>> gpurun_out/r4/c36_err.log
    python - <<PY
import json
j=json.load(open('gpurun_out/r4/c36_bench_${tag}_$i.json'))
cl={c['class'] if 'class' in c else c.get('name'): c for c in j.get('roofline_classes', [])} if isinstance(j.get('roofline_classes'), list) else j.get('roofline_classes', {})
f0=[(k, round(v.get('avg_launch_us', 0),1)) for k,v in (cl.items() if isinstance(cl, dict) else []) if 'fuse' in str(k)]
print('$tag $i headline', round(j['value'],1), round(j['ms_per_step'],3), 'f32', round(j['all_fp32_mfma']['value'],1), '|', ' '.join(str(round(o['value'],1)) for o in j['other_workloads']), '|', f0)
PY
  done
done
