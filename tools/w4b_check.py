"""wino4b (split-bf16 F(4x4,3x3)) against wino4 and the fp64 direct form; timing at the bench geometry."""
import os, sys, time
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from lfsr_amd import capi
from oracle import lfsr_oracle as O

def run(sel, *a, **k):
    if sel: os.environ["LFSR_CONV3X3"] = sel
    else: os.environ.pop("LFSR_CONV3X3", None)
    y = capi.conv3x3(*a, **k); torch.cuda.synchronize(); return y

def to_vcl(x, A):
    B, C, H, W = x.shape; h, w = H // A, W // A
    return torch.from_numpy(np.ascontiguousarray(x.reshape(B, C, h, A, w, A).transpose(0, 3, 5, 2, 4, 1))).cuda().reshape(-1, C)
def from_vcl(y, B, C, A, h, w):
    return y.cpu().numpy().reshape(B, A, A, h, w, C).transpose(0, 5, 3, 1, 4, 2).reshape(B, C, h * A, w * A)

rng = np.random.default_rng(0)
ok = True
for (B, A, h, w) in [(1, 1, 8, 32), (1, 2, 32, 32), (2, 2, 20, 37), (1, 1, 64, 96), (3, 5, 32, 32)]:
    x = rng.standard_normal((B, 64, A * h, A * w)).astype(np.float32)
    wt = (rng.standard_normal((64, 64, 3, 3)) * 0.05).astype(np.float32)
    r1 = rng.standard_normal((B, 64, A * h, A * w)).astype(np.float32)
    ref = O.leaky_relu(O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)), 0.1) + r1
    wp = capi.pack_conv_weight(torch.from_numpy(wt).cuda())
    xv, rv = to_vcl(x, A), to_vcl(r1, A)
    out = {}
    for sel in ("", "wino4b"):
        y = run(sel, xv, wp, B * A * A, h, w, slope=0.1, res1=rv)
        out[sel] = from_vcl(y, B, 64, A, h, w)
    e4, eb = np.abs(out[""] - ref).max(), np.abs(out["wino4b"] - ref).max()
    print(f"geom {(B, A, h, w)}: wino4 err {e4:.3e}  wino4b err {eb:.3e}  wino4b-wino4 {np.abs(out['wino4b'] - out['']).max():.3e}", flush=True)
    ok &= eb < 1e-4
# bench geometry
n_img, h, w = 800, 32, 32
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
r = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
wp = capi.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
ys = {}
for sel in ("", "wino4b", "halo"):
    ys[sel] = run(sel, x, wp, n_img, h, w, slope=0.1, res1=r)
print("bench geometry: wino4b - halo", float((ys["wino4b"] - ys["halo"]).abs().max()), " wino4 - halo", float((ys[""] - ys["halo"]).abs().max()), flush=True)
ok &= float((ys["wino4b"] - ys["halo"]).abs().max()) < 2e-4
# masked / two-operand variants against wino4
mk = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
for kw in (dict(slope=1.0), dict(slope=0.1, res1=r, res2=mk)):
    a = run("", x, wp, n_img, h, w, **kw); b = run("wino4b", x, wp, n_img, h, w, **kw)
    d = float((a - b).abs().max()); print("variant", sorted(kw), "wino4b - wino4", d, flush=True); ok &= d < 2e-4
wT = capi.pack_conv_weight_T(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
for kw in (dict(), dict(act=mk, act_slope=0.1), dict(act=mk, act_slope=0.1, res1=r), dict(res1=r)):
    os.environ.pop("LFSR_CONV3X3", None); a = capi.conv3x3_dgrad(x, wT, n_img, h, w, **kw); torch.cuda.synchronize()
    os.environ["LFSR_CONV3X3"] = "wino4b"; b = capi.conv3x3_dgrad(x, wT, n_img, h, w, **kw); torch.cuda.synchronize()
    d = float((a - b).abs().max()); print("dgrad variant", sorted(kw), "wino4b - wino4", d, flush=True); ok &= d < 2e-4
for rep in range(2):
    for sel in ("", "wino4b"):
        if sel: os.environ["LFSR_CONV3X3"] = sel
        else: os.environ.pop("LFSR_CONV3X3", None)
        for _ in range(5): capi.conv3x3(x, wp, n_img, h, w, slope=0.1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): capi.conv3x3(x, wp, n_img, h, w, slope=0.1)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 40 * 1e3
        e0.record()
        for _ in range(40): capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=r)
        e1.record(); torch.cuda.synchronize()
        t2 = e0.elapsed_time(e1) / 40 * 1e3
        print(f"{sel or 'wino4':8s}: {t:.1f} us plain, {t2:.1f} us with residual", flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
