"""GPU: what happens at the size limits of one launch sequence.

The kernels address their operands with 32-bit byte offsets, so the model entry points refuse a batch whose activation tensors would reach the limit
(LFSR_E_ARG) and the Python runtimes split such a batch into equal launches -- no value changes, the path is batch-invariant bit for bit.  At operator level a
row-GEMM operand past 2^31 bytes goes to the generic gather-GEMM, whose addresses are 64-bit: checked here against fp64 on a strided sample of rows."""
import json
import os

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _distg():
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.DistgSSRRuntime(5, 4)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda"))
    return rt


def test_batch_beyond_the_launch_limit_is_split_and_changes_no_bit(monkeypatch):
    rt = _distg()
    nmax = capi.max_patches_per_launch(5, 32, 32, 160, 1 << 30)
    assert nmax == 65                                            # 65 x 25 x 1024 pixels x 160 floats x 4 B < 2^30 <= 66 x ...
    B = nmax + 5
    x = torch.from_numpy(synth_input((B, 1, 160, 160), seed=11)).cuda()
    y = rt.forward(x)                                            # two launches: 65 + 5 patches
    assert y.shape == (B, 1, 640, 640) and torch.isfinite(y).all()
    for i in (0, nmax - 1, nmax, B - 1):                         # both sides of the seam
        assert torch.equal(y[i:i + 1], rt.forward(x[i:i + 1])), i
    # the C entry point itself refuses the whole batch (no silent switch to kernels that were never run at that size)
    monkeypatch.setattr(capi, "max_patches_per_launch", lambda *a, **k: 1 << 20)
    with pytest.raises(capi.LfsrError):
        rt.forward(x)


def test_linear_operand_past_2_gib_runs_on_the_64_bit_gather_gemm():
    """lfsr_linear_fwd with x and y of 2.2 GB each: the three-term bf16 row-GEMM and the fp32 row-GEMM both decline (32-bit byte offsets), the generic
    gather-GEMM computes with 64-bit addresses.  Rows sampled over the whole range, the far end included, against fp64."""
    lib = capi.load()
    M, K, N = 4_300_000, 128, 128                                # M * 128 * 4 = 2.2e9 > 2^31
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    wp = capi.pack_conv_weight(w.reshape(N, K, 1, 1))
    y = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.lfsr_linear_fwd(x.data_ptr(), K, 0, K, wp.data_ptr(), None, None, 0, 0, y.data_ptr(), N, 0, M, N, 1.0, capi.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, M, 9973, device="cuda"), torch.arange(M - 257, M, device="cuda"), torch.arange((1 << 22) - 130, (1 << 22) + 130, device="cuda")])
    ref = x[rows].double() @ w.double().t()
    err = float((y[rows].double() - ref).abs().max())
    assert err < 2e-5, err                                      # fp32 accumulation over K = 128 of O(1) products
    assert bool(torch.isfinite(y[::4099]).all())               # no row left unwritten (strided scan of the NaN pre-fill)
