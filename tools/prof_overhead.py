#!/usr/bin/env python3
"""Cost of the live hipEvent instrumentation inside bench.py's timed region: ms per DistgSSR forward (B = 32) with profiling
off, around the 3x3 conv ops only (mode 2), and around every operator class (mode 1)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
rt = capi.DistgSSRRuntime(5, 4)
rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
x = torch.from_numpy(synth_input((32, 1, 160, 160), seed=1)).cuda()
for _ in range(3): rt.forward(x)
for rep in range(2):
    for mode in (0, 2, 1):
        rt.profile(mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): rt.forward(x)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20 * 1e3
        rt.profile_read(); rt.profile(0)
        print(f"profile mode {mode}: {el:.3f} ms/step")
