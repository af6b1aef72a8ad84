#!/usr/bin/env python3
"""Times the per-view 3x3 conv op alone at the bench geometry, one line per kernel selection (LFSR_CONV3X3 = '' | wino2 | halo),
with and without a residual operand, and prints the max difference against the direct kernel.
usage: python tools/conv_time.py [n_img] [reps]"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
if os.environ.get("LFSR_LIB"): capi.LIB_PATH = os.path.abspath(os.environ["LFSR_LIB"])   # a diagnostic build (tools/build_w4_abl.sh)
capi.load()
SELS = os.environ.get("CONV_SELS", ",wino2,halo").split(",")
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 800
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
h = w = 32
M = n_img * h * w
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(M, 64, device="cuda", generator=g); r = torch.randn(M, 64, device="cuda", generator=g)
wp = capi.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
y = torch.empty(M, 64, device="cuda")
os.environ["LFSR_CONV3X3"] = "halo"
ref = capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=r).clone()
for sel in SELS:
    if sel: os.environ["LFSR_CONV3X3"] = sel
    else: os.environ.pop("LFSR_CONV3X3", None)
    for res in (None, r):
        for _ in range(5): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        err = float((y - ref).abs().max()) if res is not None else float("nan")
        print(f"conv3x3 sel={sel or 'wino4':6s} n_img={n_img} res={'y' if res is not None else 'n'}: {us:8.1f} us  {2*576*64*M/us*1e-6:7.1f} TFLOP/s(alg)  max|d vs direct| {err:.2e}", flush=True)
