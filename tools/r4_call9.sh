#!/bin/bash
# round 4, call 9: kernel trace of the headline forward: per-dispatch durations of the 53 conv ops by position in the network
set -e
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4/c9_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/c9_trace -o infer --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-split-check --steps 6 --warmup 2 > gpurun_out/r4/c9_bench.json 2> gpurun_out/r4/c9_err.log || { tail -20 gpurun_out/r4/c9_err.log; exit 1; }
ls -R gpurun_out/r4/c9_trace | head
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/r4/c9_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# forwards = sequences starting at k_initconv ending at k_head; take full-size forwards only
seqs, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "k_initconv" in n: cur = []
    if cur is not None:
        cur.append(r)
        if "k_head" in n:
            seqs.append(cur); cur = None
print(len(seqs), "forwards; launches per forward", collections.Counter(len(s) for s in seqs))
full = [s for s in seqs if len(s) == max(len(s) for s in seqs)][-6:]
pos = collections.defaultdict(list)
for s in full:
    ci = 0
    for r in s:
        n = r["Kernel_Name"]
        if "k_conv3x3_wino4" in n:
            m = re.search(r"k_conv3x3_wino4<([^>]*)>", n)
            pos[(ci, m.group(1) if m else n[:40])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            ci += 1
by_kind = collections.defaultdict(list)
for (ci, v), d in sorted(pos.items()):
    blk = ci if ci < 52 else ci
    by_kind[v].append(sum(d) / len(d))
    if ci < 12 or ci > 46: print(ci, v, round(sum(d) / len(d), 1))
for v, d in by_kind.items(): print("variant", v, "n", len(d), "avg us", round(sum(d) / len(d), 1), "min", round(min(d), 1), "max", round(max(d), 1))
# within a block: position mod 3 for the first 48 (blocks), group convs in between
import statistics
seq = [sum(d) / len(d) for (ci, v), d in sorted(pos.items())]
print("all 53:", [round(x) for x in seq])
PY
