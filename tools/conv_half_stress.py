#!/usr/bin/env python3
"""Stress of the conv's half-tile path (conv3x3_wino4.hip): many launches with fresh random data, image counts that make no / some / only half tiles; an image's
result must be the same bits whichever way its tiles were run.  usage: python tools/conv_half_stress.py [iterations]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator(device="cuda").manual_seed(123)
bad = 0
for it in range(iters):
    h, w = ((32, 32), (30, 29), (16, 32), (8, 64))[it % 4]
    tiles_per_img = ((h + 7) // 8) * ((w + 31) // 32)
    n = max(2, 400 // tiles_per_img)                       # ~400 tiles: more than one round of 256, remainder > 128 -> whole tiles only
    x = torch.randn(n * h * w, 64, device="cuda", generator=g)
    r = torch.randn(n * h * w, 64, device="cuda", generator=g)
    wt = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    wp = capi.pack_conv_weight(wt)
    whole = capi.conv3x3(x, wp, n, h, w, slope=0.1, res1=r).clone()
    for k in (max(1, 288 // tiles_per_img), max(1, 100 // tiles_per_img), max(1, 40 // tiles_per_img), 1):
        part = capi.conv3x3(x[:k * h * w], wp, k, h, w, slope=0.1, res1=r[:k * h * w])
        if not torch.equal(part, whole[:k * h * w]):
            bad += 1
            print(f"MISMATCH it={it} geometry={h}x{w} images={k}: max|d| = {float((part - whole[:k * h * w]).abs().max()):.3e}", flush=True)
torch.cuda.synchronize()
print(f"{iters} iterations, {bad} mismatches")
sys.exit(1 if bad else 0)
