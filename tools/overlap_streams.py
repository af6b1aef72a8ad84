"""Do a persistent conv launch and the EPI / Ang kernels of the same DistgSSR block overlap when issued on two streams? (B = 8 training geometry)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
B, A, h, w = int(os.environ.get("OV_B", "8")), 5, 32, 32
n_img = B * A * A; M = n_img * h * w
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, 64, device="cuda", generator=g)
wc = capi.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
we1 = capi.pack_conv_weight(torch.randn(32, 64, 1, 25, device="cuda", generator=g) * 0.05)
we2 = capi.pack_conv_weight(torch.randn(160, 32, 1, 1, device="cuda", generator=g) * 0.05)
wa1 = capi.pack_conv_weight(torch.randn(16, 64, 5, 5, device="cuda", generator=g) * 0.05)
wa2 = capi.pack_conv_weight(torch.randn(400, 16, 1, 1, device="cuda", generator=g) * 0.05)
cat = torch.empty(M, 144, device="cuda"); s1 = torch.empty(M, 64, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def spa():
    capi.conv3x3(x, wc, n_img, h, w, slope=0.1, out=s1); capi.conv3x3(s1, wc, n_img, h, w, slope=0.1, out=cat, out_choff=0)
def side():
    capi.epiconv_hv(x, we1, we2, B, A, h, w, 0.1, cat, 80, 112); capi.angconv(x, wa1, wa2, B, A, h, w, 0.1, cat, 64)
def seq():
    spa(); side()
def par():
    ev = torch.cuda.Event(); ev.record()
    sa.wait_event(ev); sb.wait_event(ev)
    with torch.cuda.stream(sa): spa()
    with torch.cuda.stream(sb): side()
    ea, eb = torch.cuda.Event(), torch.cuda.Event(); ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
def par_rev():
    ev = torch.cuda.Event(); ev.record()
    sa.wait_event(ev); sb.wait_event(ev)
    with torch.cuda.stream(sb): side()
    with torch.cuda.stream(sa): spa()
    ea, eb = torch.cuda.Event(), torch.cuda.Event(); ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
for name, fn in (("sequential", seq), ("two streams (conv first)", par), ("two streams (EPI first)", par_rev), ("sequential", seq)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} {name:28s}: {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us per (conv, conv, EPI h+v, Ang)", flush=True)
