"""time of the fused LayerNorm + feed-forward block (lfsr_ffn_ln_fwd, K1 = N2 = 128, H = 256) at the LFT scene geometry (M = 819 200 tokens) and the EPIT B = 8 geometry
(M = 204 800); LFSR_HIP_LIB selects an ablation build of ffn_b3.hip (tools/build_abl.sh, FB_ABL)"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
lib = capi.load()
K, H = 128, 256
g = torch.Generator(device="cuda").manual_seed(5)
w1 = capi.pack_conv_weight(torch.randn(H, K, 1, 1, device="cuda", generator=g) * 0.05)
w2 = capi.pack_conv_weight(torch.randn(K, H, 1, 1, device="cuda", generator=g) * 0.05)
gam, bet = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
P = capi.dev_ptr
res = []
for M in (32 * 25 * 32 * 32, 8 * 25 * 32 * 32):
    x = torch.randn(M, K, device="cuda", generator=g); y = torch.empty(M, K, device="cuda")
    def f():
        capi.check(lib.lfsr_ffn_ln_fwd(P(x), K, 0, P(gam), P(bet), 1e-5, P(w1), P(w2), P(x), K, 0, P(y), K, 0, M, K, H, K, 0.0, capi.stream_ptr()), "ffn_ln")
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(os.path.basename(os.environ.get("LFSR_HIP_LIB", "product")), "ffn_ln M=819200 %.1f us   M=204800 %.1f us" % tuple(res), flush=True)
