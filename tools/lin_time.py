"""time of the transformer linears at the LFT bench geometry (M = 32 * 25 * 32 * 32 tokens): the LayerNorm + in-projection (K = 128, N = 384) and the out-projection with
its residual (K = 128, N = 128); LFSR_HIP_LIB selects an ablation build of rowgemm_b3.hip (tools/build_abl.sh, RB_ABL)"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
lib = capi.load()
M, K = int(os.environ.get("LIN_M", str(32 * 25 * 32 * 32))), 128
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(M, K, device="cuda", generator=g)
pe = torch.randn(1024, K, device="cuda", generator=g)
win = capi.pack_conv_weight(torch.randn(384, K, 1, 1, device="cuda", generator=g) * 0.05)
wout = capi.pack_conv_weight(torch.randn(128, K, 1, 1, device="cuda", generator=g) * 0.05)
gam, bet = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
qk, v, y = torch.empty(M, 256, device="cuda"), torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
P = capi.dev_ptr
def ln():
    capi.check(lib.lfsr_linear_ln_fwd(P(x), K, 0, K, P(win), P(gam), P(bet), 1e-5, 256, P(pe), K, 1024, 1, P(qk), 256, 0, P(v), 128, 0, 256, M, 384, capi.stream_ptr()), "ln")
def out():
    capi.check(lib.lfsr_linear_fwd(P(v), K, 0, K, P(wout), None, P(x), K, 0, P(y), 128, 0, M, 128, 1.0, capi.stream_ptr()), "out")
res = []
for f in (ln, out):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(os.path.basename(os.environ.get("LFSR_HIP_LIB", "product")), "linear_ln %.1f us   out_proj %.1f us" % tuple(res), flush=True)
