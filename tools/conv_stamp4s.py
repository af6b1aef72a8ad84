#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the symmetric-wave F(4x4) conv kernel (wave 0's s_memtime deltas).  Needs a library built with
-DLFSR_CONV_DIAG; usage: LFSR_CONV3X3=wino4s python tools/conv_stamp4s.py lib.so"""
import os as _os
_os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = capi.load()
os.environ["LFSR_CONV3X3"] = "wino4s"
n_img, h, w = int(os.environ.get('N_IMG', '800')), 32, 32
M = n_img * h * w
x = torch.randn(M, 64, device="cuda"); wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wp = capi.pack_conv_weight(wt); y = torch.empty(M, 64, device="cuda"); r = torch.randn(M, 64, device="cuda")
dbg = torch.zeros(256 * 64, device="cuda")
lib.lfsr_diag_set_buffer.restype = ctypes.c_int; lib.lfsr_diag_set_buffer.argtypes = [ctypes.c_void_p]
assert lib.lfsr_diag_set_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
names = ["loop top", "VALU phase 0 (stage 2 chunks, transform, drain planes 0,1)", "MFMA phase 0 (144 MFMAs)", "VALU phase 1", "MFMA phase 1", "At M A + pair exchange"]
for res in (None, r):
    dbg.zero_()
    for _ in range(5):
        capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    d = dbg.reshape(256, 64).cpu().double()
    tot = d[:, :16].sum(1).mean()
    tiles = n_img * 4 / 256
    print(f"residual={res is not None}: {us:.1f} us per op (stamped build); mean cycles per block {tot:.0f} -> {tot / us / 1e3:.2f} GHz; {tiles:.1f} tiles per block")
    for k, nm in enumerate(names):
        print(f"   {nm:62s} {d[:, k].mean():10.0f} cyc  {100 * d[:, k].mean() / tot:5.1f}%   per tile {d[:, k].mean() / tiles:8.0f}")
