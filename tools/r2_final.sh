#!/bin/bash
export LFSR_LAB=1   # (A/B selectors of the library are live only under LFSR_LAB)
# round 2, final evidence: bench lines of the four workloads, rocprofv3 kernel-trace stats of each, FETCH_SIZE / WRITE_SIZE passes of the headline
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r2/final; mkdir -p $O; rm -rf $O/prof_* $O/pmc_*
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
for wl in epit lft train; do timeout -k 10 300 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1; done
# the same two transformer workloads with every GEMM on the fp32-MFMA kernels (the default runs the linears / FFN / tail on the bf16 pipe with exact three-term operands)
for wl in epit lft; do LFSR_ROWGEMM=f32 LFSR_FFN=f32 LFSR_UPTAIL=v2 timeout -k 10 300 python bench.py --workload $wl > $O/bench_${wl}_f32.json 2> $O/bench_${wl}_f32.err || exit 1; done
timeout -k 10 300 python tools/b3_accuracy.py > $O/b3_accuracy.log 2>&1 || exit 1
python - <<'PY'
import json
for n in ("bench", "bench_epit", "bench_lft", "bench_train", "bench_epit_f32", "bench_lft_f32"):
    j = json.loads(open(f"gpurun_out/r2/final/{n}.json").read().strip().splitlines()[-1]); print(n, round(j["value"], 1), j["unit"], round(j["ms_per_step"], 3))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_infer -o r02 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split-check > $O/prof_infer.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_epit -o r02 --output-format csv -- python3 $R/bench.py --workload epit --steps 5 --warmup 2 > $O/prof_epit.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_lft -o r02 --output-format csv -- python3 $R/bench.py --workload lft --steps 3 --warmup 1 > $O/prof_lft.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r02 --output-format csv -- python3 $R/bench.py --workload train --steps 5 --warmup 3 > $O/prof_train.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r02 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-check > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r02 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-check > $O/pmc_write.log 2>&1 || exit 1
find $O -name "*.csv" | wc -l
