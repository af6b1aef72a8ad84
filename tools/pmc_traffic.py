#!/usr/bin/env python3
"""profiles/pmc_conv3x3.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE, csv output) and the
--kernel-trace --stats pass: HBM bytes per 3x3 conv op (one launch of the Winograd kernel).
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_stats.csv> <out.json> [all_kernels.json]"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch -> value (summed over XCD instances)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    # bench.py also runs a few B = 1 forwards (its batch-split check): keep the full-size launches only (>= half the largest value of the kernel)
    out = {}
    for k, v in acc.items():
        vals = list(v.values())
        big = [x for x in vals if x >= 0.5 * max(vals)] or vals
        out[k] = (sum(big) / len(big), len(big))
    return out


fetch, write, stats, out = sys.argv[1:5]
F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
def pick(d, key):   # launch-weighted mean over every instantiation whose name contains key
    m = [v for k, v in d.items() if key in k]
    n = sum(v[1] for v in m)
    return (sum(v[0] * v[1] for v in m) / n, n) if n else (0.0, 0)
res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; bytes = "
                 "(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of "
                 "16-B-per-lane coalesced reads at 64 B; applied to this kernel's 64-B-run dword reads as well: the L2's fabric requests are the same 128-B lines); mean over the launches of the kernel "
                 "(21 of 53 ops per forward also read a residual operand)"}
total = 0.0
KEY = "k_conv3x3_wino4<false"   # forward instantiations (<MASK = false, residual operand or not, ...>)
for tag, key in (("op", KEY),):
    f, n = pick(F, key); w, _ = pick(W, key)
    b = (2.0 * f + w) * 1024.0
    res[tag] = {"kernel": key, "launches": n, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": b}
    total += b
res["kernel"] = "k_conv3x3_wino4<false, *, *>: one launch per 3x3 conv op (256 persistent 512-thread blocks over 3200 tiles of 8 x 32 pixels)"
res["hbm_bytes_per_launch"] = total
n_pix = 32 * 25 * 32 * 32
res["algorithmic_bytes_per_launch"] = n_pix * 256 * (2 + 21.0 / 53.0)   # in + out (+ residual on 21 of 53 ops)
tot_ns, calls = 0.0, 0
for r in csv.DictReader(open(stats)):
    if KEY in r["Name"]:
        tot_ns += float(r["TotalDurationNs"]); calls += int(r["Calls"])
res["rocprof_avg_us_per_op"] = tot_ns / max(calls, 1) / 1000.0
res["rocprof_avg_note"] = "mean over ALL launches of the --kernel-trace --stats pass, the B = 1 launches of bench.py's batch-split check included (2 x 53 of them per run)"
import os
trace = stats.replace("kernel_stats", "kernel_trace")
if os.path.exists(trace):   # full-size launches only (grid of the B = 32 forward), from the same pass's kernel trace
    du = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if KEY in r["Kernel_Name"]:
            du[int(r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    full = du[max(du)]
    res["rocprof_avg_us_per_full_size_op"] = sum(full) / len(full) / 1000.0
    res["rocprof_full_size_launches"] = len(full)
json.dump(res, open(out, "w"), indent=1)
if len(sys.argv) > 5:   # optional: every kernel's HBM bytes per launch (same correction) -> profiles/r01_pmc_traffic.json
    allk = {}
    for k in sorted(set(F) | set(W)):
        f, n = F.get(k, (0.0, 0)); w, _ = W.get(k, (0.0, 0))
        allk[k] = {"launches": n, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump(allk, open(sys.argv[5], "w"), indent=1)
print(json.dumps(res, indent=1))
