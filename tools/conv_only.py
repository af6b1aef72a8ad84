#!/usr/bin/env python3
"""Runs the per-view 3x3 conv op alone at the bench geometry (for rocprofv3 --pmc / --kernel-trace passes).
usage: python tools/conv_only.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
lib = capi.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_img, h, w = 800, 32, 32
M = n_img * h * w
x = torch.randn(M, 64, device="cuda"); wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wp = capi.pack_conv_weight(wt); y = torch.empty(M, 64, device="cuda")
for _ in range(reps):
    capi.conv3x3(x, wp, n_img, h, w, slope=0.1, out=y)
torch.cuda.synchronize()
print("done")
