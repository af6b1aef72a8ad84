"""Error against fp64 of the fp32-MFMA row-GEMM and of the three-term bf16 form (the default; LFSR_ROWGEMM=f32 selects the fp32-MFMA kernel) on the same data."""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
lib = capi.load()
for K, N, M, scale in ((128, 256, 65536, 1.0), (64, 128, 65536, 1.0), (128, 128, 65536, 100.0)):
    rng = np.random.default_rng(K + N)
    x = (rng.standard_normal((M, K)) * scale).astype(np.float32); w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    xd = torch.from_numpy(x).cuda(); wp = capi.pack_conv_weight(torch.from_numpy(w.reshape(N, K, 1, 1)).cuda())
    out = {}
    for sel in ("f32", ""):
        os.environ["LFSR_ROWGEMM"] = sel
        y = torch.empty(M, N, device="cuda")
        capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), None, None, 0, 0, capi.dev_ptr(y), N, 0, M, N, 1.0, capi.stream_ptr()), "linear")
        e = np.abs(y.cpu().numpy() - ref)
        out[sel or "b3"] = (e.max(), e.mean())
    t = torch.from_numpy(x).cuda() @ torch.from_numpy(w).cuda().T
    et = np.abs(t.cpu().numpy() - ref)
    print(f"K={K} N={N} scale={scale}: fp32 MFMA max {out['f32'][0]:.3e} mean {out['f32'][1]:.3e} | bf16x3 max {out['b3'][0]:.3e} mean {out['b3'][1]:.3e} | torch fp32 matmul max {et.max():.3e} mean {et.mean():.3e} | |ref| rms {np.sqrt((ref**2).mean()):.3e}")

# fused feed-forward block: y = x + W2 relu(W1 LN(x))
for K1, H, M in ((128, 256, 65536), (64, 128, 65536)):
    rng = np.random.default_rng(K1)
    x = rng.standard_normal((M, K1)).astype(np.float32); w1 = (rng.standard_normal((H, K1)) * 0.1).astype(np.float32); w2 = (rng.standard_normal((K1, H)) * 0.1).astype(np.float32)
    g = (1 + 0.3 * rng.standard_normal(K1)).astype(np.float32); b = (0.2 * rng.standard_normal(K1)).astype(np.float32)
    x64 = x.astype(np.float64)
    xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    ref = np.maximum(xn @ w1.astype(np.float64).T, 0.0) @ w2.astype(np.float64).T + x64
    xd, gd, bd = [torch.from_numpy(a).cuda() for a in (x, g, b)]
    w1p = capi.pack_conv_weight(torch.from_numpy(w1.reshape(H, K1, 1, 1)).cuda()); w2p = capi.pack_conv_weight(torch.from_numpy(w2.reshape(K1, H, 1, 1)).cuda())
    out = {}
    for sel in ("f32", ""):
        os.environ["LFSR_FFN"] = sel
        y = torch.empty(M, K1, device="cuda")
        capi.check(lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K1, 0, capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(xd), K1, 0,
                                       capi.dev_ptr(y), K1, 0, M, K1, H, K1, 0.0, capi.stream_ptr()), "ffn")
        e = np.abs(y.cpu().numpy() - ref)
        out[sel or "b3"] = (e.max(), e.mean())
    print(f"FFN K1={K1} H={H}: fp32 MFMA max {out['f32'][0]:.3e} mean {out['f32'][1]:.3e} | bf16x3 max {out['b3'][0]:.3e} mean {out['b3'][1]:.3e} | |ref| rms {np.sqrt((ref**2).mean()):.3e}")
