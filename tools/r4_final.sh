#!/bin/bash
# round 4, final evidence: the default bench line (headline + all_fp32_mfma + other_workloads + cpu_baseline), the three other workloads' own lines, rocprofv3
# kernel-trace stats of each, FETCH_SIZE / WRITE_SIZE passes of the headline (separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r4/final; mkdir -p $O; rm -rf $O/prof_* $O/pmc_*
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
for wl in epit lft train; do timeout -k 10 300 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1; done
for wl in epit lft; do timeout -k 10 300 python bench.py --workload $wl --arithmetic f32 > $O/bench_${wl}_f32.json 2> $O/bench_${wl}_f32.err || exit 1; done
python - <<'PY'
import json
for n in ("bench", "bench_epit", "bench_lft", "bench_train", "bench_epit_f32", "bench_lft_f32"):
    j = json.loads(open(f"gpurun_out/r4/final/{n}.json").read().strip().splitlines()[-1]); print(n, round(j["value"], 1), j["unit"], round(j["ms_per_step"], 3))
j = json.load(open("gpurun_out/r4/final/bench.json"))
print("all_fp32_mfma", j["all_fp32_mfma"]["value"], "roofline", j["roofline"]["frac"], j["roofline"]["avg_launch_us"])
for o in j["other_workloads"]: print("   ", o["config"][:44], round(o["value"], 1), round(o["ms_per_step"], 2), (o.get("dominant_kernel") or {}).get("operator"), (o.get("dominant_kernel") or {}).get("avg_launch_us"), (o.get("dominant_kernel") or {}).get("frac"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_infer -o r04 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split-check --no-other-workloads > $O/prof_infer.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_epit -o r04 --output-format csv -- python3 $R/bench.py --workload epit --steps 5 --warmup 2 > $O/prof_epit.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_lft -o r04 --output-format csv -- python3 $R/bench.py --workload lft --steps 3 --warmup 1 > $O/prof_lft.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r04 --output-format csv -- python3 $R/bench.py --workload train --steps 5 --warmup 3 > $O/prof_train.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r04 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-check --no-other-workloads > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r04 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-check --no-other-workloads > $O/pmc_write.log 2>&1 || exit 1
find $O -name "*.csv" | wc -l
