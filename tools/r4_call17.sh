#!/bin/bash
# round 4, call 17: forwards replayed from a HIP graph: bit-equality test, EPIT B = 1 eager vs replay
set -e
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r4/c17_tests.log 2>&1 || { tail -40 gpurun_out/r4/c17_tests.log; exit 1; }
tail -2 gpurun_out/r4/c17_tests.log
python - <<'PY'
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict
def rt_of(name):
    key = {"epit": "EPIT", "lft": "LFT"}[name]
    meta = json.load(open("tests/golden/models.json"))["models"][key]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.ModelRuntime(name, 5, 4, 5 if name == "epit" else 4, 64)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda")); return rt
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name in ("epit", "lft"):
    rt = rt_of(name); gf = capi.GraphedForward(rt)
    for B in (1, 2, 8):
        x = torch.from_numpy(synth_input((B, 1, 160, 160), seed=1)).cuda()
        print(name, "B", B, "eager ms", round(timed(lambda: rt.forward(x)), 3), "graph replay ms", round(timed(lambda: gf(x)), 3), flush=True)
PY
