#!/usr/bin/env python3
"""Throughput of the EPIT / LFT HIP forwards at the BASELINE geometry (5x5 views of 32x32, x4).
Usage: python tools/model_bench.py epit|lft [batch] [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
key = {"epit": "EPIT", "lft": "LFT"}[name]
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"][key]["full"]
sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
rt = capi.ModelRuntime(name, 5, 4, 5 if name == "epit" else 4, 64)
rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
x = torch.from_numpy(synth_input((B, 1, 160, 160), seed=1)).cuda()
for _ in range(2):
    y = rt.forward(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    y = rt.forward(x)
torch.cuda.synchronize()
el = time.perf_counter() - t0
flop = {"epit": 148.9e9, "lft": 62.4e9}[name]     # windowed accounting
print(json.dumps({"model": key, "batch": B, "patches_per_s": B * steps / el, "ms_per_step": el / steps * 1e3,
                  "model_tflops_windowed": flop * B * steps / el / 1e12}))
