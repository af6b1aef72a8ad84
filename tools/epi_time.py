"""time of both EPI passes (lfsr_epiconv_hv_fwd) at the bench geometry (B = 32, 5x5 views of 32x32); LFSR_HIP_LIB selects an ablation build (tools/build_abl.sh)"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
B, A, h, w = int(os.environ.get("EPI_B", "32")), 5, 32, 32
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B * A * A * h * w, 64, device="cuda", generator=g)
w1 = capi.pack_conv_weight(torch.randn(32, 64, 1, 25, device="cuda", generator=g) * 0.03)
w2 = capi.pack_conv_weight(torch.randn(160, 32, 1, 1, device="cuda", generator=g) * 0.15)
out = torch.zeros((x.shape[0], 144), device="cuda")
for _ in range(10): capi.epiconv_hv(x, w1, w2, B, A, h, w, 0.1, out, 80, 112)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): capi.epiconv_hv(x, w1, w2, B, A, h, w, 0.1, out, 80, 112)
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("LFSR_HIP_LIB", "product")), "%.1f us" % (e0.elapsed_time(e1) / 50 * 1e3), flush=True)
