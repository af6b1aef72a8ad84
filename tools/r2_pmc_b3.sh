#!/bin/bash
# round 2 (late): SQ counters of the three-term bf16 kernels in one EPIT forward (B = 8)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r2/pmc_b3; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/epit_$i -o r02 --output-format csv -- python3 $R/bench.py --workload epit --steps 1 --warmup 1 > $O/epit_$i.log 2>&1 || exit 1
done
python3 - <<'P'
import csv, glob, collections, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
keys = ("k_ffn_b3<128", "k_rowgemm_b3<128, true", "k_rowgemm_b3<128, false", "k_rowgemm_b3<64", "k_up_tail3", "k_epi_attn_mfma")
for f in sorted(glob.glob(f"{R}/gpurun_out/r2/pmc_b3/epit_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]: acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (k, cn), d in acc.items():
        v = list(d.values())
        out.setdefault(k, {})[cn] = sum(v) / len(v)
json.dump(out, open(f"{R}/gpurun_out/r2/pmc_b3/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
