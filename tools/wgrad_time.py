import os, sys, torch
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
sys.path.insert(0, "/root/repo")
from lfsr_amd import capi
n_img, h, w = 200, 32, 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n_img * h * w, 64, device="cuda", generator=g); dy = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
out = []
for sel in ("", "direct", ""):
    if sel: os.environ["LFSR_WGRAD3"] = sel
    else: os.environ.pop("LFSR_WGRAD3", None)
    for _ in range(10): capi.conv3x3_wgrad(dy, x, n_img, h, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): capi.conv3x3_wgrad(dy, x, n_img, h, w)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{sel or 'wino'} {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (wgrad + reduce + workspace alloc)")
print(os.path.basename(os.environ.get("LFSR_HIP_LIB", "product")), " | ".join(out))
