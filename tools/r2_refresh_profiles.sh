#!/bin/bash
# copy the evidence of tools/r2_final.sh (gpurun_out/r2/final) into profiles/r02_* and rebuild profiles/r02_pmc_traffic.json
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/r2/final
cp $F/bench.json profiles/r02_bench.json; cp $F/bench_epit.json profiles/r02_epit_bench.json; cp $F/bench_lft.json profiles/r02_lft_bench.json; cp $F/bench_train.json profiles/r02_bench_train.json
cp $F/bench_epit_f32.json profiles/r02_epit_bench_f32mfma.json; cp $F/bench_lft_f32.json profiles/r02_lft_bench_f32mfma.json; cp $F/b3_accuracy.log profiles/r02_logs/b3_accuracy.log
cp $F/prof_infer/r02_kernel_stats.csv profiles/r02_kernel_stats.csv; cp $F/prof_epit/r02_kernel_stats.csv profiles/r02_epit_kernel_stats.csv
cp $F/prof_lft/r02_kernel_stats.csv profiles/r02_lft_kernel_stats.csv; cp $F/prof_train/r02_kernel_stats.csv profiles/r02_train_kernel_stats.csv
python tools/pmc_traffic.py $F/pmc_fetch/r02_counter_collection.csv $F/pmc_write/r02_counter_collection.csv $F/prof_infer/r02_kernel_stats.csv /tmp/r02_pmc_conv.json /tmp/r02_pmc_all.json > /dev/null
python - <<'PY'
import json
conv = json.load(open('/tmp/r02_pmc_conv.json')); allk = json.load(open('/tmp/r02_pmc_all.json'))
json.dump({"conv3x3": conv, "hbm_bytes_per_launch": conv["hbm_bytes_per_launch"], "all_kernels": allk}, open('profiles/r02_pmc_traffic.json', 'w'), indent=1)
print("conv3x3: %.0f MB per launch = %.3f x algorithmic; rocprof avg %.1f us" % (conv["hbm_bytes_per_launch"] / 1e6, conv["hbm_bytes_per_launch"] / conv["algorithmic_bytes_per_launch"], conv["rocprof_avg_us_per_full_size_op"]))
for n in ("bench", "epit_bench", "lft_bench", "bench_train", "epit_bench_f32mfma", "lft_bench_f32mfma"):
    j = json.loads(open(f"profiles/r02_{n}.json").read().strip().splitlines()[-1]); print(n, round(j["value"], 1), j["unit"], round(j["ms_per_step"], 3))
PY
