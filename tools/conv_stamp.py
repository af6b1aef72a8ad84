#!/usr/bin/env python3
"""Diagnostic: where a persistent conv3x3 block spends its cycles (wave 0's s_memtime deltas per segment).
Needs a library built with -DLFSR_CONV_DIAG; usage: python tools/conv_stamp.py lib.so"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = capi.load()
n_img, h, w = int(os.environ.get('N_IMG', '800')), 32, 32
M = n_img * h * w
x = torch.randn(M, 64, device="cuda"); wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wp = capi.pack_conv_weight(wt); y = torch.empty(M, 64, device="cuda"); r = torch.randn(M, 64, device="cuda")
dbg = torch.zeros(256 * 64, device="cuda")
import ctypes
lib.lfsr_diag_set_buffer.restype = ctypes.c_int; lib.lfsr_diag_set_buffer.argtypes = [ctypes.c_void_p]
assert lib.lfsr_diag_set_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0   # the DIAG build's own stamp-buffer argument (never an operand slot)
for res in (None, r):
    dbg.zero_()
    for _ in range(5):
        capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=res, out=y)
    torch.cuda.synchronize()
    sel = os.environ.get("LFSR_CONV3X3", "")
    if sel == "":   # F(4x4,3x3) kernel (default): per (chunk, stage) segments, chunk barriers, the epilogue at the head of a pass
        d = dbg.reshape(256, 64).cpu().double()
        names = (["chunk %d MFMA stream after A" % k for k in range(4)] + ["-"] * 4 + ["At M A + exchange writes"] + ["-"] + ["B1 wait after chunk %d" % c for c in range(4)] + ["-"] * 2
                 + ["barrier after chunk %d" % c for c in range(4)] + ["exchange barrier"] + ["-"] * 3
                 + ["chunk %d MFMA stream before A" % c for c in range(4)] + ["chunk %d barrier A wait" % c for c in range(4)])
        for c in range(4):
            names += [f"P step {c}: halo registers -> LDS (+ operand request)", f"P step {c}: barrier A wait", f"P step {c}: patch reads + next halo request",
                      f"P step {c}: input transform + V writes", f"P step {c}: drain a plane", f"P step {c}: chunk barrier wait (BURST: B1 wait)", f"P step {c}: (BURST) B wait", "-"]
        names[62] = "P: exchange barrier wait"
        tot = d[:, :32].sum(1).mean()
        print(f"residual={res is not None}: mean cycles per block {tot:.0f} ({n_img * 4 / 256:.1f} tiles per block); producer total {d[:, 32:].sum(1).mean():.0f}")
        for k in range(64):
            if names[k] == '-': continue
            print(f"   {names[k]:28s} {d[:, k].mean():12.0f} cyc  {100 * d[:, k].mean() / tot:5.1f}%   per tile {d[:, k].mean() / (n_img * 4 / 256):9.0f}")
        continue
    wino = sel[:1] != "h"
    d = dbg.reshape(256, 32).cpu().double() if wino else dbg.reshape(-1)[:2048].reshape(256, 8).cpu().double()
    if os.environ.get("LFSR_CONV3X3", "")[:1] == "h":
        names = ["9 taps", "seam barrier wait", "transpose+stores", "post-epilogue barrier", "halo LDS write+barrier"]
    else:   # Winograd kernel (default)
        names = ["-", "seam barrier wait", "round 0 write+barrier", "round 0 read/store+barrier", "round 1 write+barrier",
                 "round 1 read/store+barrier", "ring restart+barrier"]
    nk = len(names)
    if wino:
        names = names + ["-"] + ["unit %2d" % u for u in range(16)]
        nk = len(names)
    tot = d[:, :nk].sum(1).mean()
    print(f"residual={res is not None}: mean cycles per block {tot:.0f} (tiles per block ~12)")
    for k in range(nk):
        print(f"   {names[k]:28s} {d[:, k].mean():12.0f} cyc  {100 * d[:, k].mean() / tot:5.1f}%   per tile {d[:, k].mean() / 12:9.0f}")
