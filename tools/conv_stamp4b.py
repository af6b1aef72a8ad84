#!/usr/bin/env python3
"""Diagnostic: cycles per segment of the split-bf16 F(4x4) conv kernel (s_memtime deltas of every consumer and producer wave).  Needs a
library built with -DLFSR_CONV_DIAG (tools/build_w4b_var.sh "diag:-DLFSR_CONV_DIAG"); usage: python tools/conv_stamp4b.py lib.so"""
import os as _os
_os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = capi.load()
os.environ["LFSR_CONV3X3"] = "wino4b"
n_img, h, w = int(os.environ.get('N_IMG', '800')), 32, 32
M = n_img * h * w
x = torch.randn(M, 64, device="cuda"); wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wp = capi.pack_conv_weight(wt); y = torch.empty(M, 64, device="cuda"); r = torch.randn(M, 64, device="cuda")
dbg = torch.zeros(256 * 64, device="cuda")
lib.lfsr_diag_set_buffer.restype = ctypes.c_int; lib.lfsr_diag_set_buffer.argtypes = [ctypes.c_void_p]
assert lib.lfsr_diag_set_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
cn = ["MFMAs of the lower half", "MFMAs of the upper half", "wait at H1", "wait at H2", "At M A + plane writes", "wait at E"]
pn = ["X: pending V write, stage 1 + 2, operand request", "wait at E", "wait at H1", "Y: halo -> LDS, split, V write", "Y: drain a plane, halo request", "wait at H2"]
for res in (None, r):
    for _ in range(5): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    d = dbg.reshape(256, 64).cpu().double()   # (every launch overwrites its block's 64 floats: the last launch)
    tiles = n_img * 4 / 256
    tot = d[:, 0:6].sum(1).mean()
    print(f"residual={res is not None}: {us:.1f} us per op (stamped build); consumer wave 0 cycles per block {tot:.0f} -> {tot / us / 1e3:.2f} GHz; {tiles:.1f} tiles per block")
    for wv in range(4):
        print(f"  consumer wave {wv}: " + "  ".join(f"{cn[k]} {d[:, 8 * wv + k].mean() / tiles:7.0f}" for k in range(6)) + f"   | sum {d[:, 8 * wv:8 * wv + 6].sum(1).mean() / tiles:7.0f} per tile")
    for wv in range(4):
        print(f"  producer wave {wv} (xi half {wv >> 1}, nu half {wv & 1}): " + "  ".join(f"{pn[k]} {d[:, 32 + 8 * wv + k].mean() / tiles:7.0f}" for k in range(6)) + f"   | sum {d[:, 32 + 8 * wv:32 + 8 * wv + 6].sum(1).mean() / tiles:7.0f}")
