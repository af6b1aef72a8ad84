#!/bin/bash
# round 4: SQ / traffic counters of the forward's kernels (headline bench, B = 32) and of the LFT window attention, separate rocprofv3 --pmc passes (no trace domains besides the kernel trace)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r4/pmc; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/infer_$i -o r04 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --no-split-check --steps 2 --warmup 1 > $O/infer_$i.log 2>&1 || exit 1
  if [ $i -ge 6 ] || [ $i -eq 1 ]; then
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/lft_$i -o r04 --output-format csv -- python3 $R/bench.py --workload lft --steps 1 --warmup 1 > $O/lft_$i.log 2>&1 || exit 1
  fi
done
python3 - <<'P'
import csv, glob, collections, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
for tag, keys in (("infer", ("k_epi_b3", "k_conv3x3_wino4<false, false", "k_conv3x3_wino4<false, true", "k_rowgemm_b3", "k_ang_fused", "k_epi_wino5")), ("lft", ("k_win_attn_mfma", "k_ffn_b3", "k_rowgemm_b3", "k_up_tail4"))):
    for f in sorted(glob.glob(f"{R}/gpurun_out/r4/pmc/{tag}_*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            for k in keys:
                if k in r["Kernel_Name"]: acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, cn), d in acc.items():
            v = sorted(d.values()); big = [x for x in v if x >= 0.5 * v[-1]] or v
            out.setdefault(tag + ":" + k, {})[cn] = sum(big) / len(big)
    # kernel durations from the same passes' traces (full-size launches)
    for f in sorted(glob.glob(f"{R}/gpurun_out/r4/pmc/{tag}_1/**/*kernel_trace.csv", recursive=True)):
        du = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k in keys:
                if k in r["Kernel_Name"]: du[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in du.items():
            v.sort(); big = [x for x in v if x >= 0.5 * v[-1]]
            out.setdefault(tag + ":" + k, {})["avg_us_under_pmc"] = sum(big) / len(big) / 1e3
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch (2 x FETCH_SIZE + WRITE_SIZE) x 1024"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
json.dump(out, open(f"{R}/gpurun_out/r4/pmc/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
