#!/bin/bash
# copy the evidence of tools/r4_final.sh (gpurun_out/r4/final) into profiles/r04_* and rebuild profiles/r04_pmc_traffic.json
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/r4/final
cp $F/bench.json profiles/r04_bench.json; cp $F/bench_epit.json profiles/r04_epit_bench.json; cp $F/bench_lft.json profiles/r04_lft_bench.json; cp $F/bench_train.json profiles/r04_bench_train.json
cp $F/bench_epit_f32.json profiles/r04_epit_bench_f32mfma.json; cp $F/bench_lft_f32.json profiles/r04_lft_bench_f32mfma.json
f() { find $F/$1 -name "*$2" | head -1; }
cp "$(f prof_infer kernel_stats.csv)" profiles/r04_kernel_stats.csv; cp "$(f prof_epit kernel_stats.csv)" profiles/r04_epit_kernel_stats.csv
cp "$(f prof_lft kernel_stats.csv)" profiles/r04_lft_kernel_stats.csv; cp "$(f prof_train kernel_stats.csv)" profiles/r04_train_kernel_stats.csv
python tools/pmc_traffic.py "$(f pmc_fetch counter_collection.csv)" "$(f pmc_write counter_collection.csv)" "$(f prof_infer kernel_stats.csv)" /tmp/r04_pmc_conv.json /tmp/r04_pmc_all.json > /dev/null
python - <<'PY'
import json
conv = json.load(open('/tmp/r04_pmc_conv.json')); allk = json.load(open('/tmp/r04_pmc_all.json'))
json.dump({"conv3x3": conv, "hbm_bytes_per_launch": conv["hbm_bytes_per_launch"], "all_kernels": allk}, open('profiles/r04_pmc_traffic.json', 'w'), indent=1)
print("conv3x3: %.0f MB per launch = %.3f x algorithmic; rocprof avg %.1f us" % (conv["hbm_bytes_per_launch"] / 1e6, conv["hbm_bytes_per_launch"] / conv["algorithmic_bytes_per_launch"], conv.get("rocprof_avg_us_per_full_size_op", conv["rocprof_avg_us_per_op"])))
for n in ("bench", "epit_bench", "lft_bench", "bench_train", "epit_bench_f32mfma", "lft_bench_f32mfma"):
    j = json.loads(open(f"profiles/r04_{n}.json").read().strip().splitlines()[-1]); print(n, round(j["value"], 1), j["unit"], round(j["ms_per_step"], 3))
PY
