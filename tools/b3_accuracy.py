"""Error against fp64 of the fp32-MFMA row-GEMM and of the three-term bf16 form (the default; LFSR_ROWGEMM=f32 selects the fp32-MFMA kernel) on the same data."""
import os, sys
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
