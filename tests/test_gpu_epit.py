"""GPU: EPIT operator classes and whole forward (config 'EPIT 5x5 x4 inference') through the C ABI vs the numpy oracle
(fp64, reference formulation) and the reference's golden outputs."""
import sys

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.synth import synth_input
from oracle import lfsr_oracle as O
from tests.helpers import model_case, psnr

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def test_layernorm():
    lib = capi.load()
    for C in (64, 128):
        x = rnd((1000, C), 1, 2.0)
        g, b = rnd((C,), 2), rnd((C,), 3)
        y = torch.empty(1000, C, device="cuda")
        xd, gd, bd = dev(x), dev(g), dev(b)          # keep the device tensors alive across the asynchronous call
        capi.check(lib.lfsr_layernorm_fwd(capi.dev_ptr(xd), C, 0, None, 0, 0, 1, capi.dev_ptr(gd), capi.dev_ptr(bd), capi.dev_ptr(y), C, 0, 1000, C, 1e-5,
                                          capi.stream_ptr()), "ln")
        ref = O.layer_norm(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64))
        assert np.abs(y.cpu().numpy() - ref).max() < 1e-5


@pytest.mark.parametrize("M", [777, 5003])      # 777: gather-GEMM; 5003 (ragged against 64- and 128-row tiles): the row-streaming GEMM, 64x64 and 128x128 tiles
@pytest.mark.parametrize("cin,N,slope", [(64, 128, 1.0), (128, 256, 0.0), (256, 128, 1.0), (128, 64, 0.2), (128, 96, 1.0), (128, 128, 1.0), (144, 64, 0.1)])
def test_linear(cin, N, slope, M, monkeypatch):
    if cin == 144 and M < 2048:
        pytest.skip("K = 144 (fuse.0) exists in the row-streaming GEMM only; lfsr_pointwise_fwd covers it at small M")
    lib = capi.load()
    x, w, r, b = rnd((M, cin), 4), rnd((N, cin), 5, 0.1), rnd((M, N), 6), rnd((N,), 7)
    xd, rd, bd, wp = dev(x), dev(r), dev(b), capi.pack_conv_weight(dev(w.reshape(N, cin, 1, 1)))
    z = x.astype(np.float64) @ w.astype(np.float64).T
    for sel in ("", "f32", "128"):      # default: exact three-term bf16 operands on the bf16 MFMA pipe where K = 64 / 128 and there is no bias; f32: the fp32-MFMA form; 128: its 128 x 128 tiles
        monkeypatch.setenv("LFSR_ROWGEMM", sel)
        for use_r, use_b in ((True, False), (False, True), (False, False)):
            y = torch.full((M, N), float("nan"), device="cuda")
            capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), cin, 0, cin, capi.dev_ptr(wp), capi.dev_ptr(bd) if use_b else None,
                                           capi.dev_ptr(rd) if use_r else None, N, 0, capi.dev_ptr(y), N, 0, M, N, slope, capi.stream_ptr()), "linear")
            zz = z + (b if use_b else 0.0)
            ref = np.where(zz >= 0, zz, zz * slope) + (r if use_r else 0.0)
            assert np.abs(y.cpu().numpy() - ref).max() < ATOL, (sel, use_r, use_b)


@pytest.mark.parametrize("K1,H,M", [(128, 256, 777), (64, 128, 2100), (128, 256, 70000)])
def test_ffn_fused(K1, H, M):
    """y = res + W2 relu(W1 x): the fused feed-forward block (EPIT.py:84-90,126; LFT.py:151-156,216-221) against fp64 numpy"""
    lib = capi.load()
    x, w1, w2, r = rnd((M, K1), 14), rnd((H, K1), 15, 0.1), rnd((K1, H), 16, 0.1), rnd((M, K1), 17)
    y = torch.empty(M, K1, device="cuda")
    xd, rd = dev(x), dev(r)
    w1p = capi.pack_conv_weight(dev(w1.reshape(H, K1, 1, 1)))
    w2p = capi.pack_conv_weight(dev(w2.reshape(K1, H, 1, 1)))
    capi.check(lib.lfsr_ffn_fwd(capi.dev_ptr(xd), K1, 0, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(rd), K1, 0,
                                capi.dev_ptr(y), K1, 0, M, K1, H, K1, 0.0, capi.stream_ptr()), "ffn")
    hid = np.maximum(x.astype(np.float64) @ w1.astype(np.float64).T, 0.0)
    ref = hid @ w2.astype(np.float64).T + r
    assert np.abs(y.cpu().numpy() - ref).max() < ATOL
    # unsupported shapes are refused, not silently mis-computed
    assert lib.lfsr_ffn_fwd(capi.dev_ptr(xd), K1, 0, capi.dev_ptr(w1p), capi.dev_ptr(w2p), None, 0, 0,
                            capi.dev_ptr(y), K1, 0, M, K1, H + 8, K1, 0.0, capi.stream_ptr()) == -1


@pytest.mark.parametrize("K1,H,M", [(128, 256, 777), (64, 128, 2100), (128, 256, 70000)])
def test_ffn_with_layernorm_inside(K1, H, M):
    """y = res + W2 relu(W1 LayerNorm(x)) with the norm formed in registers (feed_forward.0 ... .4) against fp64 numpy; rows with a large common offset"""
    lib = capi.load()
    x, w1, w2 = rnd((M, K1), 24) + 3.0 * rnd((M, 1), 25), rnd((H, K1), 15, 0.1), rnd((K1, H), 16, 0.1)
    g, b = (1.0 + 0.3 * rnd((K1,), 26)).astype(np.float32), rnd((K1,), 27, 0.2)
    y = torch.empty(M, K1, device="cuda")
    xd, gd, bd = dev(x), dev(g), dev(b)
    w1p = capi.pack_conv_weight(dev(w1.reshape(H, K1, 1, 1)))
    w2p = capi.pack_conv_weight(dev(w2.reshape(K1, H, 1, 1)))
    capi.check(lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K1, 0, capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(xd), K1, 0,
                                   capi.dev_ptr(y), K1, 0, M, K1, H, K1, 0.0, capi.stream_ptr()), "ffn_ln")
    x64 = x.astype(np.float64)
    xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    ref = np.maximum(xn @ w1.astype(np.float64).T, 0.0) @ w2.astype(np.float64).T + x64
    assert np.abs(y.cpu().numpy() - ref).max() < ATOL
    assert lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K1, 0, None, capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), None, 0, 0,
                               capi.dev_ptr(y), K1, 0, M, K1, H, K1, 0.0, capi.stream_ptr()) == -1


@pytest.mark.parametrize("K,M,with_pe", [(128, 5003, False), (128, 4096, True), (64, 5003, True), (64, 100, False)])
@pytest.mark.parametrize("form", ["f32", "bf16x3"])
def test_linear_with_layernorm_inside(K, M, with_pe, form, monkeypatch):
    """q | k from LayerNorm(x + pe), v from x in one launch against fp64; the fp32-MFMA form returns the same bits as lfsr_layernorm_fwd + two lfsr_linear_fwd"""
    if form == "f32":
        monkeypatch.setenv("LFSR_ROWGEMM", "f32")
    else:
        monkeypatch.delenv("LFSR_ROWGEMM", raising=False)
    lib = capi.load()
    N, split = 3 * K, 2 * K
    x, w = rnd((M, K), 31) + 2.0 * rnd((M, 1), 32), rnd((N, K), 33, 0.1)
    g, b = (1.0 + 0.3 * rnd((K,), 34)).astype(np.float32), rnd((K,), 35, 0.2)
    pe_rows, pe_div = 7, 3
    pe = rnd((pe_rows, K), 36)
    xd, gd, bd, ped = dev(x), dev(g), dev(b), dev(pe)
    wp = capi.pack_conv_weight(dev(w.reshape(N, K, 1, 1)))
    qk, v = torch.empty(M, split, device="cuda"), torch.empty(M, N - split, device="cuda")
    capi.check(lib.lfsr_linear_ln_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, split,
                                      capi.dev_ptr(ped) if with_pe else None, K, pe_rows, pe_div, capi.dev_ptr(qk), split, 0, capi.dev_ptr(v), N - split, 0, split,
                                      M, N, capi.stream_ptr()), "linear_ln")
    xn = torch.empty(M, K, device="cuda")
    capi.check(lib.lfsr_layernorm_fwd(capi.dev_ptr(xd), K, 0, capi.dev_ptr(ped) if with_pe else None, K, pe_rows, pe_div, capi.dev_ptr(gd), capi.dev_ptr(bd),
                                      capi.dev_ptr(xn), K, 0, M, K, 1e-5, capi.stream_ptr()), "ln")
    x64 = x.astype(np.float64) + (pe[(np.arange(M) // pe_div) % pe_rows] if with_pe else 0.0)
    ref_n = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    assert np.abs(qk.cpu().numpy() - ref_n @ w[:split].astype(np.float64).T).max() < ATOL
    assert np.abs(v.cpu().numpy() - x.astype(np.float64) @ w[split:].astype(np.float64).T).max() < ATOL
    if M >= 2048 and form == "f32":      # (below that lfsr_linear_fwd runs the gather-GEMM, a different summation order)
        qk2, v2 = torch.empty_like(qk), torch.empty_like(v)
        capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xn), K, 0, K, capi.dev_ptr(wp), None, None, 0, 0, capi.dev_ptr(qk2), split, 0, M, split, 1.0, capi.stream_ptr()), "qk")
        capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp[split * K:]), None, None, 0, 0, capi.dev_ptr(v2), N - split, 0, M, N - split, 1.0,
                                       capi.stream_ptr()), "v")
        assert torch.equal(qk, qk2) and torch.equal(v, v2)


@pytest.mark.parametrize("vertical", [0, 1])
@pytest.mark.parametrize("geom", [(2, 3, 6, 8), (1, 5, 32, 32), (1, 5, 20, 32)])
@pytest.mark.parametrize("path", ["mfma", "valu"])
def test_epi_attention_vs_masked_mha(vertical, geom, path, monkeypatch):
    """window predicate == the reference's additive -inf mask (EPIT.py:93-108) inside nn.MultiheadAttention's core; both the MFMA kernel
    (attn_mfma.hip, default) and the VALU kernels, at a reduced geometry and at EPIT's own (5 x 32 = 160 tokens per sequence; 5 x 20: ragged last tile)"""
    lib = capi.load()
    if path == "valu":
        monkeypatch.setenv("LFSR_ATTN", "valu")
    else:
        monkeypatch.delenv("LFSR_ATTN", raising=False)
    (B, A, h, w), E, NH = geom, 128, 8
    npix = B * A * A * h * w
    q, k, v = rnd((npix, E), 7), rnd((npix, E), 8), rnd((npix, E), 9)
    o = torch.empty(npix, E, device="cuda")
    HW = h * w
    if not vertical:
        args = (B, A, w, A * A * HW, HW, 1, A, h, A * HW, w)
    else:
        args = (B, A, h, A * A * HW, A * HW, w, A, w, HW, 1)
    qd, kd, vd = dev(q), dev(k), dev(v)
    capi.check(lib.lfsr_window_attn_fwd(capi.dev_ptr(qd), E, 0, capi.dev_ptr(kd), E, 0, capi.dev_ptr(vd), E, 0, capi.dev_ptr(o), E, 0, NH, E // NH,
                                        *args, A, A, 5, 6, 0, capi.stream_ptr()), "attn")
    # reference: tokens (L, N, E) in the rearranged order of AltFilter.forward (EPIT.py:150/156)
    def to_seq(t):
        t = t.astype(np.float64).reshape(B, A, A, h, w, E)                                  # b u v y x e
        if not vertical:
            return t.transpose(1, 3, 0, 2, 4, 5).reshape(A * h, B * A * w, E)               # (u y) (b v x)
        return t.transpose(2, 4, 0, 1, 3, 5).reshape(A * w, B * A * h, E)                   # (v x) (b u y)
    L = A * (w if vertical else h)
    mask = O.epit_gen_mask(A, w if vertical else h, 2 * A, 11, np.float64)
    Q, K, V = to_seq(q), to_seq(k), to_seq(v)
    assert Q.shape[0] == L
    hd = E // NH
    Qh = Q.reshape(L, -1, hd).transpose(1, 0, 2)
    Kh = K.reshape(L, -1, hd).transpose(1, 0, 2)
    Vh = V.reshape(L, -1, hd).transpose(1, 0, 2)
    S = Qh @ Kh.transpose(0, 2, 1) / np.sqrt(hd) + mask
    Pm = np.exp(S - S.max(-1, keepdims=True))
    Pm /= Pm.sum(-1, keepdims=True)
    ref_seq = (Pm @ Vh).transpose(1, 0, 2).reshape(L, -1, E)
    got = to_seq(o.cpu().numpy())
    assert np.abs(got - ref_seq).max() < 1e-5


TAGS = ["a5h8s4", "a3h6w8s2"]


def runtime(case, sd):
    rt = capi.ModelRuntime("epit", case["A"], case["s"], 5, 64)
    rt.load_state([(k, dev(v)) for k, v in sd.items()], torch.device("cuda", 0))
    return rt


@pytest.mark.parametrize("ln_fuse", ["default", "0", "1"])   # default: all norms inside the consuming kernels / every norm its own launch / only the feed-forward norm fused
@pytest.mark.parametrize("tag", TAGS)
def test_epit_small_vs_golden_and_oracle(tag, ln_fuse, monkeypatch):
    if ln_fuse == "default":
        monkeypatch.delenv("LFSR_LN_FUSE", raising=False)
    else:
        monkeypatch.setenv("LFSR_LN_FUSE", ln_fuse)
    case, sd, x, npz = model_case("EPIT", tag)
    y = runtime(case, sd).forward(dev(x)).cpu().numpy()
    gold = npz[tag + "_out"]
    ref = O.epit_forward(x, sd, case["A"], case["s"])
    assert np.abs(y - ref).max() < ATOL
    assert np.abs(y - gold).max() < ATOL
    assert psnr(y, gold) >= 80.0


def test_epit_full_patch():
    case, sd, x1, npz = model_case("EPIT", "full")
    x = np.concatenate([x1, synth_input(x1.shape, seed=5)], axis=0)
    rt = runtime(case, sd)
    y = rt.forward(dev(x)).cpu().numpy()
    assert y.shape == (2, 1, 640, 640)
    assert np.abs(y[:1, :, ::8, ::8] - npz["full_sample"]).max() < ATOL          # the reference itself
    ref = O.epit_forward(x[:1], sd, 5, 4)
    assert np.abs(y[:1] - ref).max() < ATOL
    label = synth_input(ref.shape, seed=2)
    assert abs(psnr(y[:1], label) - psnr(ref, label)) <= 0.01
    assert np.array_equal(rt.forward(dev(x[1:])).cpu().numpy(), y[1:])            # batch independence


def test_epit_plugin_surface():
    import importlib
    from argparse import Namespace
    sys.path.insert(0, capi._HERE)
    try:
        M = importlib.import_module("model.SR.EPIT")
    finally:
        sys.path.remove(capi._HERE)
    case, sd, x, npz = model_case("EPIT", "a3h6w8s2")
    net = M.get_model(Namespace(angRes_in=3, angRes_out=3, scale_factor=2))
    assert [k for k in net.state_dict()] == [k for k, _ in case["spec"]]
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to("cuda:0").eval()
    with torch.no_grad():
        y = net(dev(x), [3, 3])
    assert np.abs(y.cpu().numpy() - npz["a3h6w8s2_out"]).max() < ATOL
