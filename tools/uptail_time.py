#!/usr/bin/env python3
"""Time lfsr_up_tail_fwd (s = 4, A = 5, 32 x 32 views) at several batch sizes in one process: us per launch and per patch.  Timing only (random operands)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi

lib = capi.load()
A, h, w, s = 5, 32, 32, 4
g = torch.Generator(device="cuda").manual_seed(1)
w0 = torch.randn(64 * s * s, 64, 1, 1, device="cuda", generator=g) * 0.05
w0p = capi.pack_conv_weight(w0, perm=1, ch=64) if "perm" in capi.pack_conv_weight.__code__.co_varnames else capi.pack_conv_weight(w0)
w3 = torch.randn(64 * 9, device="cuda", generator=g) * 0.05
st = capi.stream_ptr()
for rounds in range(2):
    for B in [int(b) for b in (sys.argv[1:] or ["1", "8", "16", "32", "64"])]:
        f = torch.randn(B * A * A * h * w, 64, device="cuda", generator=g)
        x = torch.randn(B, 1, A * h, A * w, device="cuda", generator=g)
        out = torch.empty(B, 1, A * h * s, A * w * s, device="cuda")
        def run():
            rc = lib.lfsr_up_tail_fwd(capi.dev_ptr(f), 64, 0, capi.dev_ptr(w0p), capi.dev_ptr(w3), capi.dev_ptr(x), capi.dev_ptr(out), B, A, h, w, s, 0.1, st)
            assert rc == 0, rc
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"B={B:3d}  {us:8.1f} us per launch  {us / B:7.2f} us per patch  ({B * 200} blocks = {B * 200 / 256:.2f} rounds of 256)", flush=True)
