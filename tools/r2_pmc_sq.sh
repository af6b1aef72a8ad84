#!/bin/bash
# round 2: SQ counters of the 3x3 conv op alone (tools/conv_only.py), the Winograd-form weight gradient and the EPI kernels (train step)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r2/pmc_sq; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/conv_$i -o r02 --output-format csv -- python3 $R/tools/conv_only.py 3 > $O/conv_$i.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $O/train_$i -o r02 --output-format csv -- python3 $R/bench.py --workload train --steps 1 --warmup 1 > $O/train_$i.log 2>&1 || exit 1
done
python3 - <<'P'
import csv, glob, collections, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
for tag, keys in (("conv", ("k_conv3x3_wino4<false, false",)), ("train", ("k_wgrad_conv3_wino", "k_wgrad_epi0_lines", "k_epi0_dgrad_lines", "k_epi_wino5"))):
    for f in sorted(glob.glob(f"{R}/gpurun_out/r2/pmc_sq/{tag}_*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            for k in keys:
                if k in r["Kernel_Name"]: acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, cn), d in acc.items():
            v = sorted(d.values()); big = [x for x in v if x >= 0.5 * v[-1]] or v
            out.setdefault(k, {})[cn] = sum(big) / len(big)
json.dump(out, open(f"{R}/gpurun_out/r2/pmc_sq/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
