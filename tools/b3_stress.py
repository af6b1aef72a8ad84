"""Random-shape sweep of the three-term bf16 kernels against fp64 (row-GEMM, LN + q|k|v projection, fused FFN with LayerNorm): ragged M, both K, residual / no residual."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfsr_amd import capi
lib = capi.load()
rng = np.random.default_rng(2026)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
worst = {"linear": 0.0, "linear_ln": 0.0, "ffn_ln": 0.0}
ncase = 0
for it in range(40):
    K = int(rng.choice([64, 128])); M = int(rng.integers(2048, 40000)); N = int(rng.choice([64, 128, 192, 256, 384]))
    x = (rng.standard_normal((M, K)) + rng.standard_normal((M, 1))).astype(np.float32); w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32); use_r = bool(rng.integers(0, 2))
    xd, rd = dev(x), dev(r); wp = capi.pack_conv_weight(dev(w.reshape(N, K, 1, 1)))
    y = torch.full((M, N), float("nan"), device="cuda")
    capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), None, capi.dev_ptr(rd) if use_r else None, N, 0, capi.dev_ptr(y), N, 0, M, N, 1.0, capi.stream_ptr()), "linear")
    ref = x.astype(np.float64) @ w.astype(np.float64).T + (r if use_r else 0.0)
    worst["linear"] = max(worst["linear"], float(np.abs(y.cpu().numpy() - ref).max())); ncase += 1
    # LN + projection (N = 3 K, q|k normalised, v raw)
    N3, split = 3 * K, 2 * K
    w3 = (rng.standard_normal((N3, K)) * 0.1).astype(np.float32); g = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32); b = (0.2 * rng.standard_normal(K)).astype(np.float32)
    with_pe = bool(rng.integers(0, 2)); pe_rows, pe_div = int(rng.integers(1, 40)), int(rng.integers(1, 9)); pe = rng.standard_normal((pe_rows, K)).astype(np.float32)
    w3p = capi.pack_conv_weight(dev(w3.reshape(N3, K, 1, 1))); gd, bd, ped = dev(g), dev(b), dev(pe)
    qk, v = torch.full((M, split), float("nan"), device="cuda"), torch.full((M, N3 - split), float("nan"), device="cuda")
    capi.check(lib.lfsr_linear_ln_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(w3p), capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, split, capi.dev_ptr(ped) if with_pe else None, K, pe_rows, pe_div,
                                      capi.dev_ptr(qk), split, 0, capi.dev_ptr(v), N3 - split, 0, split, M, N3, capi.stream_ptr()), "linear_ln")
    x64 = x.astype(np.float64) + (pe[(np.arange(M) // pe_div) % pe_rows] if with_pe else 0.0)
    xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    e = max(float(np.abs(qk.cpu().numpy() - xn @ w3[:split].astype(np.float64).T).max()), float(np.abs(v.cpu().numpy() - x.astype(np.float64) @ w3[split:].astype(np.float64).T).max()))
    worst["linear_ln"] = max(worst["linear_ln"], e); ncase += 1
    # fused FFN with LayerNorm
    H = 2 * K
    w1 = (rng.standard_normal((H, K)) * 0.1).astype(np.float32); w2 = (rng.standard_normal((K, H)) * 0.1).astype(np.float32)
    w1p = capi.pack_conv_weight(dev(w1.reshape(H, K, 1, 1))); w2p = capi.pack_conv_weight(dev(w2.reshape(K, H, 1, 1)))
    yf = torch.full((M, K), float("nan"), device="cuda")
    capi.check(lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K, 0, capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(xd), K, 0,
                                   capi.dev_ptr(yf), K, 0, M, K, H, K, 0.0, capi.stream_ptr()), "ffn_ln")
    x64r = x.astype(np.float64)
    xnr = (x64r - x64r.mean(-1, keepdims=True)) / np.sqrt(x64r.var(-1, keepdims=True) + 1e-5) * g + b
    reff = np.maximum(xnr @ w1.astype(np.float64).T, 0.0) @ w2.astype(np.float64).T + x64r
    worst["ffn_ln"] = max(worst["ffn_ln"], float(np.abs(yf.cpu().numpy() - reff).max())); ncase += 1
print(f"{ncase} cases; worst |hip - fp64|: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
assert all(v < 1e-4 for v in worst.values())
