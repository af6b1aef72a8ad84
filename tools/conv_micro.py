#!/usr/bin/env python3
"""Micro-benchmark of lfsr_conv3x3_fwd alone at the bench geometry (B=32 -> 800 view images of 32x32).
Usage: python tools/conv_micro.py [lib.so]   (env LFSR_CONV_DBG / LFSR_CONV3X3 select diagnostic variants)"""
import ctypes as C
import os
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi

if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = capi.load()
n_img, h, w = 800, 32, 32
M = n_img * h * w
x = torch.randn(M, 64, device="cuda")
wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wp = capi.pack_conv_weight(wt)
y = torch.empty(M, 64, device="cuda")
r = torch.randn(M, 64, device="cuda")
flop = 2.0 * 576 * 64 * M
for res in (None, r):
    for _ in range(3):
        capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    n = 20
    e0.record()
    for _ in range(n):
        capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"dbg={os.environ.get('LFSR_CONV_DBG','0')} sel={os.environ.get('LFSR_CONV3X3','halo')} residual={res is not None}: "
          f"{ms*1e3:.1f} us  {flop/ms/1e9:.1f} TFLOP/s  ({flop/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak)")
