#!/usr/bin/env python3
"""HBM roofline check of the stand-alone index kernels (a1-a7): algorithmic bytes (one read + one write of every element)
/ hip-event time, against ~8 TB/s peak (6.3 TB/s is what a float4 copy reaches, MI355X_MICROARCH.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def report(name, nbytes, t):
    print(f"{name:44s} {nbytes/1e6:9.1f} MB  {t*1e6:9.1f} us  {nbytes/t/1e12:6.2f} TB/s  ({nbytes/t/8e12*100:5.1f}% of 8 TB/s)")


x = torch.rand(32, 64, 160, 160, device="cuda")
report("copy_ (torch, reference point)", 2 * x.numel() * 4, timeit(lambda: x.clone()))
report("sai2macpi  (32,64,160,160)", 2 * x.numel() * 4, timeit(lambda: capi.sai2macpi(x, 5)))
report("macpi2sai  (32,64,160,160)", 2 * x.numel() * 4, timeit(lambda: capi.macpi2sai(x, 5)))
x1 = torch.rand(32, 1, 160, 160, device="cuda")
report("sai2macpi  (32,1,160,160)", 2 * x1.numel() * 4, timeit(lambda: capi.sai2macpi(x1, 5)))
p = torch.rand(4, 1024, 160, 160, device="cuda")
report("pixel_shuffle2d r=4 (4,1024,160,160)", 2 * p.numel() * 4, timeit(lambda: capi.pixel_shuffle2d(p, 4)))
p5 = torch.rand(32, 400, 32, 32, device="cuda")
report("pixel_shuffle2d r=5 (32,400,32,32)", 2 * p5.numel() * 4, timeit(lambda: capi.pixel_shuffle2d(p5, 5)))
q = torch.rand(32, 160, 160, 32, device="cuda")
report("pixel_shuffle1d f=5 (32,160,160,32)", 2 * q.numel() * 4, timeit(lambda: capi.pixel_shuffle1d(q, 5)))
lr = torch.rand(5 * 512, 5 * 512, device="cuda")
sub = capi.lf_divide(lr, 5, 32, 16)
report("lf_divide 5x5x512x512 -> 32x32 patches", (lr.numel() + sub.numel()) * 4, timeit(lambda: capi.lf_divide(lr, 5, 32, 16)))
big = torch.rand(8, 8, 640, 640, device="cuda")
o = capi.lf_integrate(big, 5, 128, 64, 512, 512)
report("lf_integrate 64 x 640^2 -> 5x5x512x512", 2 * o.numel() * 4, timeit(lambda: capi.lf_integrate(big, 5, 128, 64, 512, 512)))
v = torch.rand(32 * 25600, 64, device="cuda")
report("vcl_to_nchw (32,64,160,160)", 2 * v.numel() * 4, timeit(lambda: capi.vcl_to_nchw(v, 32, 64, 5, 32, 32, 1)))
