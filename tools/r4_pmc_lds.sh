#!/bin/bash
# round 4: LDS bank-conflict share and matrix-pipe busy cycles of the EPIT / LFT kernels (separate rocprofv3 --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r4/pmc_lds; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for wl in epit lft; do
  i=0
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/${wl}_$i -o r04 --output-format csv -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 > $O/${wl}_$i.log 2>&1 || exit 1
  done
done
python3 - <<'P'
import csv, glob, collections, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
for wl in ("epit", "lft"):
    for f in sorted(glob.glob(f"{R}/gpurun_out/r4/pmc_lds/{wl}_*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, cn), d in acc.items():
            v = sorted(d.values()); big = [x for x in v if x >= 0.5 * v[-1]] or v
            out.setdefault(wl + ":" + k, {})[cn] = sum(big) / len(big)
for k, d in sorted(out.items()):
    if d.get("SQ_LDS_IDX_ACTIVE", 0) > 1e5 or d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 1e6:
        conf = d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1)
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(d.get("SQ_BUSY_CYCLES", 1) / 8 * 1.0, 1)
        print(f"{k:70s} lds_conflict_share {conf:5.2f}  lds_active/CU {d.get('SQ_LDS_IDX_ACTIVE',0)/256:10.0f}  mfma_busy/SIMD {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024:10.0f}  valu/wave? {d.get('SQ_INSTS_VALU',0):.3g}  wave_cycles {d.get('SQ_WAVE_CYCLES',0):.3g}")
json.dump(out, open(f"{R}/gpurun_out/r4/pmc_lds/summary.json", "w"), indent=1)
P
