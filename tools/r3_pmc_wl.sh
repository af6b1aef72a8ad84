#!/bin/bash
# SQ counters of a workload's kernels: tools/r3_pmc_wl.sh <workload> <tag>   (separate rocprofv3 --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
WL=$1; TAG=$2
O=$R/gpurun_out/r3/pmc_$TAG; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/p_$i -o r03 --output-format csv -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 > $O/p_$i.log 2>&1 || { tail -5 $O/p_$i.log; exit 1; }
done
python3 - $O <<'P'
import csv, glob, collections, json, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob(f"{O}/p_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
du = collections.defaultdict(list)
for f in sorted(glob.glob(f"{O}/p_1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        du[r["Kernel_Name"].split("(")[0][-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, cs in acc.items():
    d = {}
    for cn, disp in cs.items():
        v = sorted(disp.values()); big = [x for x in v if x >= 0.5 * v[-1]] or v
        d[cn] = sum(big) / len(big)
    v = sorted(du.get(k, [0])); big = [x for x in v if x >= 0.5 * v[-1]] or v
    d["avg_us"] = sum(big) / len(big) / 1e3; d["launches"] = len(v); d["total_ms"] = sum(v) / 1e6
    out[k] = d
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
    us = d["avg_us"]; wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: {us:.1f} us x {d['launches']}  MFMA busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/(us*1800+1e-9):.2f} | wait_any {d.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} valu_active {d.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} wait_lds {d.get('SQ_WAIT_INST_LDS',0)/wc:.2f} | LDS conflict {d.get('SQ_LDS_BANK_CONFLICT',0)/(d.get('SQ_LDS_IDX_ACTIVE',0) or 1):.2f} | VALU {d.get('SQ_INSTS_VALU',0):.3g} LDS {d.get('SQ_INSTS_LDS',0):.3g} | HBM {(2*d.get('FETCH_SIZE',0)+d.get('WRITE_SIZE',0))*1024/1e6:.0f} MB")
P
