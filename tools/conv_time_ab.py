import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import torch
sys.path.insert(0, "/root/repo")
from lfsr_amd import capi
if os.environ.get("W4B_LIB"): capi.LIB_PATH = os.environ["W4B_LIB"]
n_img, h, w = 800, 32, 32
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
r = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
wp = capi.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
res = []
for sel in (os.environ.get("W4B_SELS", "wino4b").split(",")):
    if sel and sel != "wino4": os.environ["LFSR_CONV3X3"] = sel
    else: os.environ.pop("LFSR_CONV3X3", None)
    for _ in range(5): capi.conv3x3(x, wp, n_img, h, w, slope=0.1)
    torch.cuda.synchronize()
    ts = []
    for kw in ({}, {"res1": r}):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    res.append(f"{sel}: {ts[0]:.1f} / {ts[1]:.1f} us")
print(os.path.basename(os.environ.get("W4B_LIB", "product")), " | ".join(res), flush=True)
