"""GPU: DistgSSR operator classes and the whole forward through the C ABI against the numpy oracle
(reference formulation, fp64) and the golden vectors produced by the reference itself.

Tolerance (BASELINE north_star: fp32 conv path within 0.01 dB PSNR): we hold the sharper gates
max|err| <= 1e-4 and PSNR(hip, ref) >= 80 dB and check |dPSNR| <= 0.01 dB against a label."""
import os
import sys

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_tensor
from oracle import lfsr_oracle as O
from tests.helpers import model_case, psnr

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def to_vcl(x_macpi, A):
    return capi.nchw_to_vcl(dev(x_macpi), A, 1)


def from_vcl(v, B, C, A, h, w, choff=0):
    return capi.vcl_to_nchw(v, B, C, A, h, w, 1, choff).cpu().numpy()


GEOMS = [(1, 5, 8, 8), (2, 3, 6, 8), (1, 5, 32, 32), (3, 2, 5, 7)]
WIDE_GEOMS = [(1, 2, 20, 40), (2, 1, 37, 70)]   # view images wider than one 32-column tile, ragged in both directions


@pytest.mark.parametrize("B,A,h,w", GEOMS)
def test_conv3x3(B, A, h, w):
    x = rnd((B, 64, A * h, A * w), 1)
    wt = rnd((64, 64, 3, 3), 2, 0.05)
    r1 = rnd((B, 64, A * h, A * w), 3)
    ref = O.leaky_relu(O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)), 0.1)
    wp = capi.pack_conv_weight(dev(wt))
    y = capi.conv3x3(to_vcl(x, A), wp, B * A * A, h, w, slope=0.1)
    assert np.abs(from_vcl(y, B, 64, A, h, w) - ref).max() < ATOL
    # residual form: conv + r1 + r1 (two residual inputs), no activation
    ref2 = O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)) + 2 * r1
    rv = to_vcl(r1, A)
    y2 = capi.conv3x3(to_vcl(x, A), wp, B * A * A, h, w, slope=1.0, res1=rv, res2=rv)
    assert np.abs(from_vcl(y2, B, 64, A, h, w) - ref2).max() < ATOL
    # a lone residual passed as the SECOND operand (the kernel treats it as its first)
    y3 = capi.conv3x3(to_vcl(x, A), wp, B * A * A, h, w, slope=1.0, res2=rv)
    assert np.abs(from_vcl(y3, B, 64, A, h, w) - (ref2 - r1)).max() < ATOL


@pytest.mark.parametrize("B,A,h,w", WIDE_GEOMS)
def test_conv3x3_wide_views(B, A, h, w, monkeypatch):
    """several tile columns per view image (the persistent kernels walk contiguous tile ranges across them), all kernel selections"""
    x = rnd((B, 64, A * h, A * w), 31)
    wt = rnd((64, 64, 3, 3), 32, 0.05)
    r1 = rnd((B, 64, A * h, A * w), 33)
    ref = O.leaky_relu(O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)), 0.1) + r1
    wp = capi.pack_conv_weight(dev(wt))
    xv, rv = to_vcl(x, A), to_vcl(r1, A)
    for sel in ("", "wino2", "halo"):
        if sel: monkeypatch.setenv("LFSR_CONV3X3", sel)
        else: monkeypatch.delenv("LFSR_CONV3X3", raising=False)
        y = capi.conv3x3(xv, wp, B * A * A, h, w, slope=0.1, res1=rv)
        torch.cuda.synchronize()
        assert np.abs(from_vcl(y, B, 64, A, h, w) - ref).max() < ATOL, sel
    monkeypatch.delenv("LFSR_CONV3X3", raising=False)


def test_packed_conv_weight_carries_winograd_copy():
    """lfsr_pack_conv_weight(64,64,3,3) = direct [9][64][64] pack followed by U = G g Gt in the fragment order of the kernel"""
    wt = rnd((64, 64, 3, 3), 21, 0.05)
    wp = capi.pack_conv_weight(dev(wt)).cpu().numpy()
    assert wp.size == (9 + 16 + 36) * 64 * 64
    direct = wt.reshape(64, 64, 9).transpose(2, 0, 1).reshape(-1)           # [tap][n][k]
    assert np.array_equal(wp[:9 * 64 * 64], direct)
    ref = O.winograd_pack(wt)
    w2 = wp[9 * 64 * 64:25 * 64 * 64]
    assert np.abs(w2 - ref).max() <= 1e-9 and np.mean(w2 == ref) > 0.999   # fp64 compute, one rounding
    ref4 = O.winograd4_pack(wt)                                             # F(4x4,3x3) copy (conv3x3_wino4.hip)
    w4 = wp[25 * 64 * 64:61 * 64 * 64]
    assert np.abs(w4 - ref4).max() <= 1e-8 and np.mean(w4 == ref4) > 0.99


def test_conv3x3_tail_and_fallback_kernels(monkeypatch):
    """320 tiles = one full round of 256 CUs + 64 leftover tiles -> the channel-split tail launch (two blocks per tile);
    the direct 9-tap kernel (LFSR_CONV3X3=halo), its tail form (LFSR_CONV_TAIL=halo) and the v1 gather-GEMM must agree too"""
    B, A, h, w = 5, 4, 32, 32
    x = rnd((B, 64, A * h, A * w), 11)
    wt = rnd((64, 64, 3, 3), 12, 0.05)
    r1 = rnd((B, 64, A * h, A * w), 13)
    ref = O.leaky_relu(O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)), 0.1) + r1
    wp = capi.pack_conv_weight(dev(wt))
    xv, rv = to_vcl(x, A), to_vcl(r1, A)
    for env in ({}, {"LFSR_CONV3X3": "wino2"}, {"LFSR_CONV3X3": "wino2", "LFSR_CONV_TAIL": "halo"}, {"LFSR_CONV3X3": "wino2", "LFSR_CONV_NOTAIL": "1"},
                {"LFSR_CONV3X3": "halo"}, {"LFSR_CONV3X3": "gather"}):
        for k in ("LFSR_CONV_TAIL", "LFSR_CONV_NOTAIL", "LFSR_CONV3X3"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        y = capi.conv3x3(xv, wp, B * A * A, h, w, slope=0.1, res1=rv)
        torch.cuda.synchronize()
        assert np.abs(from_vcl(y, B, 64, A, h, w) - ref).max() < ATOL, env


def test_conv3x3_bench_size_properties(monkeypatch):
    """BASELINE geometry of the headline bench (800 view images of 32x32: 12 full rounds of 256 tiles + 128 leftover tiles), where
    the fp64 oracle is too slow: the Winograd kernel must agree with the direct 9-tap kernel, and be linear in its input"""
    n_img, h, w = 32 * 25, 32, 32
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
    z = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
    r = torch.randn(n_img * h * w, 64, device="cuda", generator=g)
    wp = capi.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05)
    for k in ("LFSR_CONV_TAIL", "LFSR_CONV_NOTAIL", "LFSR_CONV3X3"):
        monkeypatch.delenv(k, raising=False)
    yw = capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=r)
    monkeypatch.setenv("LFSR_CONV3X3", "halo")
    yd = capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=r)
    monkeypatch.delenv("LFSR_CONV3X3")
    torch.cuda.synchronize()
    assert float((yw - yd).abs().max()) < 2e-4                      # F(4x4,3x3) vs direct on unit-variance data (|y| up to ~6)
    monkeypatch.setenv("LFSR_CONV3X3", "wino2")
    y2 = capi.conv3x3(x, wp, n_img, h, w, slope=0.1, res1=r)
    monkeypatch.delenv("LFSR_CONV3X3")
    torch.cuda.synchronize()
    assert float((y2 - yd).abs().max()) < 2e-5                      # F(2x2,3x3): two fp32 evaluation orders of the same sums
    # linearity (slope 1, no residual): conv(2 x - 3 z) == 2 conv(x) - 3 conv(z)
    lin = capi.conv3x3(2.0 * x - 3.0 * z, wp, n_img, h, w, slope=1.0)
    ref = 2.0 * capi.conv3x3(x, wp, n_img, h, w, slope=1.0) - 3.0 * capi.conv3x3(z, wp, n_img, h, w, slope=1.0)
    torch.cuda.synchronize()
    assert float((lin - ref).abs().max()) < 2e-5 * float(ref.abs().max())     # (F(4x4,3x3): ~1e-5 relative)


@pytest.mark.parametrize("h,w", [(32, 32), (30, 29)])
def test_conv3x3_half_tiles_change_no_bit(h, w):
    """The launch's last partial round runs as HALF tiles when it would fill at most half of the CUs (conv3x3_wino4.hip: two CUs per tile, the 36 transform positions
    split between waves, partial inverse transforms summed through LDS).  Which tiles are split depends on the image count only: on 256 CUs, 100 images (400 tiles) run
    whole, 72 images (288 tiles) run images 64..71 as halves, 25 images (100 tiles) run every tile as halves -- an image's result must not change by a bit
    (forward with activation, two-residual form, data gradient with the LeakyReLU' mask and a residual; aligned and ragged geometry)."""
    g = torch.Generator(device="cuda").manual_seed(17)
    n = 100
    x = torch.randn(n * h * w, 64, device="cuda", generator=g)
    r = torch.randn(n * h * w, 64, device="cuda", generator=g)
    act = torch.randn(n * h * w, 64, device="cuda", generator=g)
    wt = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    wp, wtp = capi.pack_conv_weight(wt), capi.pack_conv_weight_T(wt)
    forms = [lambda k: capi.conv3x3(x[:k * h * w], wp, k, h, w, slope=0.1),
             lambda k: capi.conv3x3(x[:k * h * w], wp, k, h, w, slope=1.0, res1=r[:k * h * w], res2=r[:k * h * w]),
             lambda k: capi.conv3x3_dgrad(x[:k * h * w], wtp, k, h, w, res1=r[:k * h * w], act=act[:k * h * w], act_slope=0.1)]
    for f in forms:
        whole = f(100).clone()
        for k in (72, 25):
            part = f(k)
            assert torch.equal(part, whole[:k * h * w]), (k, h, w)
    # ... and the half-tile launch against the fp64 oracle
    xs = x[:25 * h * w].reshape(25, h, w, 64).permute(0, 3, 1, 2).cpu().numpy().astype(np.float64)
    ref = O.leaky_relu(O.conv2d(xs, wt.cpu().numpy().astype(np.float64), padding=(1, 1)), 0.1)
    y = capi.conv3x3(x[:25 * h * w], wp, 25, h, w, slope=0.1).reshape(25, h, w, 64).permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(y - ref).max() < ATOL


@pytest.mark.parametrize("B,A,h,w", GEOMS)
def test_angconv(B, A, h, w):
    x = rnd((B, 64, A * h, A * w), 4)
    w1 = rnd((16, 64, A, A), 5, 0.03)
    w2 = rnd((A * A * 16, 16, 1, 1), 6, 0.2)
    x64 = x.astype(np.float64)
    t = O.leaky_relu(O.conv2d(x64, w1.astype(np.float64), stride=(A, A)), 0.1)
    ref = O.pixel_shuffle(O.leaky_relu(O.conv2d(t, w2.astype(np.float64)), 0.1), A)
    out = torch.zeros((B * A * A * h * w, 144), device="cuda")
    capi.angconv(to_vcl(x, A), capi.pack_conv_weight(dev(w1)), capi.pack_conv_weight(dev(w2), perm=1, ch=16), B, A, h, w, 0.1, out, 64)
    assert np.abs(from_vcl(out, B, 16, A, h, w, choff=64) - ref).max() < ATOL
    assert float(out[:, :64].abs().max()) == 0.0 and float(out[:, 80:].abs().max()) == 0.0   # wrote only its slice


@pytest.mark.parametrize("vertical", [0, 1])
@pytest.mark.parametrize("B,A,h,w", GEOMS)
def test_epiconv(B, A, h, w, vertical):
    x = rnd((B, 64, A * h, A * w), 7)
    w1 = rnd((32, 64, 1, A * A), 8, 0.03)
    w2 = rnd((A * 32, 32, 1, 1), 9, 0.15)

    def epi(t):
        e = O.leaky_relu(O.conv2d(t, w1.astype(np.float64), stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        return O.pixel_shuffle1d(O.leaky_relu(O.conv2d(e, w2.astype(np.float64)), 0.1), A)
    x64 = x.astype(np.float64)
    ref = epi(np.ascontiguousarray(x64.transpose(0, 1, 3, 2))).transpose(0, 1, 3, 2) if vertical else epi(x64)
    out = torch.zeros((B * A * A * h * w, 144), device="cuda")
    capi.epiconv(to_vcl(x, A), capi.pack_conv_weight(dev(w1)), capi.pack_conv_weight(dev(w2)), B, A, h, w, vertical, 0.1, out, 112)
    assert np.abs(from_vcl(out, B, 32, A, h, w, choff=112) - ref).max() < ATOL


@pytest.mark.parametrize("B,A,h,w", GEOMS)
def test_epiconv_hv_one_launch(B, A, h, w):
    x = rnd((B, 64, A * h, A * w), 27)
    w1 = rnd((32, 64, 1, A * A), 28, 0.03)
    w2 = rnd((A * 32, 32, 1, 1), 29, 0.15)

    def epi(t):
        e = O.leaky_relu(O.conv2d(t, w1.astype(np.float64), stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        return O.pixel_shuffle1d(O.leaky_relu(O.conv2d(e, w2.astype(np.float64)), 0.1), A)
    x64 = x.astype(np.float64)
    refh = epi(x64)
    refv = epi(np.ascontiguousarray(x64.transpose(0, 1, 3, 2))).transpose(0, 1, 3, 2)
    out = torch.zeros((B * A * A * h * w, 144), device="cuda")
    capi.epiconv_hv(to_vcl(x, A), capi.pack_conv_weight(dev(w1)), capi.pack_conv_weight(dev(w2)), B, A, h, w, 0.1, out, 80, 112)
    assert np.abs(from_vcl(out, B, 32, A, h, w, choff=80) - refh).max() < ATOL
    assert np.abs(from_vcl(out, B, 32, A, h, w, choff=112) - refv).max() < ATOL
    assert float(out[:, :80].abs().max()) == 0.0


@pytest.mark.parametrize("cin,N", [(144, 64), (64, 64), (64, 40), (32, 160), (16, 400)])
def test_pointwise(cin, N):
    M = 1000
    x = rnd((M, cin), 10)
    wt = rnd((N, cin, 1, 1), 11, 0.1)
    b = rnd((N,), 12)
    ref = O.leaky_relu(x.astype(np.float64) @ wt.reshape(N, cin).astype(np.float64).T + b, 0.1)
    y = capi.pointwise(dev(x), cin, capi.pack_conv_weight(dev(wt)), N, slope=0.1, bias=dev(b))
    assert np.abs(y.cpu().numpy() - ref).max() < ATOL


@pytest.mark.parametrize("B,A,h,w", GEOMS)
def test_initconv(B, A, h, w):
    x = synth_input((B, 1, A * h, A * w), 1)
    wt = rnd((64, 1, 3, 3), 13, 0.3)
    ref = O.conv2d(O.sai2macpi(x.astype(np.float64), A), wt.astype(np.float64), dilation=(A, A), padding=(A, A))
    y = capi.initconv(dev(x), dev(wt), A)
    assert np.abs(from_vcl(y, B, 64, A, h, w) - ref).max() < 1e-5


@pytest.mark.parametrize("s", [2, 4])
@pytest.mark.parametrize("B,A,h,w", GEOMS[:3])
def test_upsample_head(B, A, h, w, s):
    f = rnd((B, 64, A * h, A * w), 14)           # MacPI features
    x = synth_input((B, 1, A * h, A * w), 1)
    w0 = rnd((64 * s * s, 64, 1, 1), 15, 0.1)
    b0 = rnd((64 * s * s,), 16, 0.1)
    w2 = rnd((1, 64, 1, 1), 17, 0.1)
    f64 = f.astype(np.float64)
    up = O.conv2d(O.pixel_shuffle(O.conv2d(O.macpi2sai(f64, A), w0.astype(np.float64), b0.astype(np.float64)), s), w2.astype(np.float64))
    ref = up + O.interp_bilinear(x.astype(np.float64), s)
    y = capi.upsample_head(to_vcl(f, A), dev(w0), dev(b0), dev(w2), dev(x), A, s)
    assert np.abs(y.cpu().numpy() - ref).max() < ATOL


# ---- whole model ------------------------------------------------------------------------------------

def build_runtime(case, sd):
    rt = capi.DistgSSRRuntime(case["A"], case["s"])
    rt.load_state([(k, dev(v)) for k, v in sd.items()], torch.device("cuda", 0))
    return rt


TAPS = ["init_conv", "b0_out", "g0_out", "disentg_out"]


@pytest.mark.parametrize("tag", ["a5h8s4", "a3h6w8s2"])
def test_distgssr_small_vs_golden_and_oracle(tag):
    case, sd, x, npz = model_case("DistgSSR", tag)
    rt = build_runtime(case, sd)
    y, bufs = rt.forward(dev(x), taps=[True] * 5)
    y = y.cpu().numpy()
    otaps = {}
    ref = O.distgssr_forward(x, sd, case["A"], case["s"], taps=otaps)
    for i, name in enumerate(TAPS):
        assert np.abs(bufs[i].cpu().numpy() - otaps[name]).max() < ATOL, name
    cat = np.concatenate([otaps["b0_spa"], otaps["b0_ang"], otaps["b0_epih"], otaps["b0_epiv_t"].transpose(0, 1, 3, 2)], axis=1)
    assert np.abs(bufs[4].cpu().numpy() - cat).max() < ATOL
    gold = npz[tag + "_out"]
    assert np.abs(y - ref).max() < ATOL
    assert np.abs(y - gold).max() < ATOL
    assert psnr(y, gold) >= 80.0


def test_distgssr_full_patch():
    """BASELINE config geometry (5x5, 32x32 -> 128x128, x4) at B=2 against the oracle and the reference's
    checksummed full-size output (strided sample)."""
    case, sd, x1, npz = model_case("DistgSSR", "full")
    x = np.concatenate([x1, synth_input(x1.shape, seed=5)], axis=0)
    rt = build_runtime(case, sd)
    y = rt.forward(dev(x)).cpu().numpy()
    assert y.shape == (2, 1, 640, 640)
    assert np.abs(y[:1, :, ::8, ::8] - npz["full_sample"]).max() < ATOL        # the reference itself
    ref = O.distgssr_forward(x, sd, 5, 4, dtype=np.float64)
    assert np.abs(y - ref).max() < ATOL
    assert psnr(y, ref) >= 80.0
    label = synth_input(y.shape, seed=2)
    assert abs(psnr(y, label) - psnr(ref, label)) <= 0.01
    # batch independence: patch 0 alone gives the same bits as patch 0 inside the batch
    y0 = rt.forward(dev(x[:1])).cpu().numpy()
    assert np.array_equal(y0, y[:1])


def test_distgssr_plugin_surface():
    """model/SR/DistgSSR.py mirror: get_model / get_loss / weights_init, state_dict round trip, no-grad forward."""
    import importlib
    from argparse import Namespace
    sys.path.insert(0, capi._HERE)
    try:
        M = importlib.import_module("model.SR.DistgSSR")
    finally:
        sys.path.remove(capi._HERE)
    case, sd, x, npz = model_case("DistgSSR", "a3h6w8s2")
    net = M.get_model(Namespace(angRes_in=3, angRes_out=3, scale_factor=2))
    net.apply(M.weights_init)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to("cuda:0").eval()
    with torch.no_grad():
        y = net(dev(x), [3, 3])
    assert np.abs(y.cpu().numpy() - npz["a3h6w8s2_out"]).max() < ATOL
    with pytest.raises(capi.LfsrError):
        with torch.no_grad():
            net(torch.from_numpy(x), [3, 3])          # CPU tensor: no fallback
    loss = M.get_loss(None)(y, torch.zeros_like(y), [3, 3])
    assert loss.ndim == 0
    # weights changed in place -> repacked
    with torch.no_grad():
        net.upsample[2].weight.mul_(0.0)
        y2 = net(dev(x), [3, 3])
    ref_skip = O.interp_bilinear(x.astype(np.float64), 2)
    assert np.abs(y2.cpu().numpy() - ref_skip).max() < 1e-5


# ---- geometries beyond one 32-column tile / one fused-EPI line: the generic paths ----------------------------------------

@pytest.mark.parametrize("B,A,h,w", [(1, 3, 40, 48), (1, 5, 33, 36)])
def test_ops_large_views(B, A, h, w):
    """views wider than the 32-column conv tile (several tiles per image) and longer than a fused EPI line (gather-GEMM fallback)"""
    x = rnd((B, 64, A * h, A * w), 31)
    wt = rnd((64, 64, 3, 3), 32, 0.05)
    ref = O.leaky_relu(O.conv2d(x.astype(np.float64), wt.astype(np.float64), dilation=(A, A), padding=(A, A)), 0.1)
    y = capi.conv3x3(to_vcl(x, A), capi.pack_conv_weight(dev(wt)), B * A * A, h, w, slope=0.1)
    assert np.abs(from_vcl(y, B, 64, A, h, w) - ref).max() < ATOL
    w1 = rnd((32, 64, 1, A * A), 33, 0.03)
    w2 = rnd((A * 32, 32, 1, 1), 34, 0.15)

    def epi(t):
        e = O.leaky_relu(O.conv2d(t, w1.astype(np.float64), stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        return O.pixel_shuffle1d(O.leaky_relu(O.conv2d(e, w2.astype(np.float64)), 0.1), A)
    x64 = x.astype(np.float64)
    out = torch.zeros((B * A * A * h * w, 144), device="cuda")
    capi.epiconv_hv(to_vcl(x, A), capi.pack_conv_weight(dev(w1)), capi.pack_conv_weight(dev(w2)), B, A, h, w, 0.1, out, 80, 112)
    assert np.abs(from_vcl(out, B, 32, A, h, w, choff=80) - epi(x64)).max() < ATOL
    assert np.abs(from_vcl(out, B, 32, A, h, w, choff=112) - epi(np.ascontiguousarray(x64.transpose(0, 1, 3, 2))).transpose(0, 1, 3, 2)).max() < ATOL


def test_distgssr_large_view_patch():
    """whole forward on 40x36 views (not the 32x32 the kernels are tuned for) against the oracle"""
    case, sd, _, _ = model_case("DistgSSR", "a3h6w8s2")
    A, s, h, w = 3, 2, 40, 36
    x = synth_input((1, 1, A * h, A * w), seed=11)
    rt = build_runtime(case, sd)
    y = rt.forward(dev(x)).cpu().numpy()
    ref = O.distgssr_forward(x, sd, A, s)
    assert np.abs(y - ref).max() < ATOL


@pytest.mark.parametrize("A,s,h,w,B", [(7, 2, 6, 6, 1), (5, 3, 8, 8, 2), (2, 2, 6, 6, 1)])
def test_distgssr_other_angres_and_scales(A, s, h, w, B):
    """angular resolutions beyond the fused-EPI fast path (A = 7: gather-GEMM EPI; A = 2: even A, general column arithmetic of
    the 1xA^2 conv) and scale 3, against the oracle (weights regenerated for the matching state_dict shapes)"""
    from lfsr_amd.synth import synth_state_dict
    AA = A * A
    spec = [("init_conv.weight", (64, 1, 3, 3))]
    for g in range(4):
        for b in range(4):
            p = f"disentg.Group.{g}.Block.{b}."
            spec += [(p + "SpaConv.0.weight", (64, 64, 3, 3)), (p + "SpaConv.2.weight", (64, 64, 3, 3)),
                     (p + "AngConv.0.weight", (16, 64, A, A)), (p + "AngConv.2.weight", (AA * 16, 16, 1, 1)),
                     (p + "EPIConv.0.weight", (32, 64, 1, AA)), (p + "EPIConv.2.weight", (A * 32, 32, 1, 1)),
                     (p + "fuse.0.weight", (64, 144, 1, 1)), (p + "fuse.2.weight", (64, 64, 3, 3))]
        spec.append((f"disentg.Group.{g}.conv.weight", (64, 64, 3, 3)))
    spec += [("disentg.conv.weight", (64, 64, 3, 3)), ("upsample.0.weight", (64 * s * s, 64, 1, 1)), ("upsample.0.bias", (64 * s * s,)),
             ("upsample.2.weight", (1, 64, 1, 1))]
    sd = synth_state_dict(spec, seed=3)
    x = synth_input((B, 1, A * h, A * w), seed=7)
    rt = capi.DistgSSRRuntime(A, s)
    rt.load_state([(k, dev(v)) for k, v in sd.items()], torch.device("cuda", 0))
    y = rt.forward(dev(x)).cpu().numpy()
    ref = O.distgssr_forward(x, sd, A, s)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < ATOL


def test_batch32_equals_single_patches():
    """configs[1] at its own batch: the B = 32 forward equals the same 32 patches run one at a time, bit for bit (tile scheduling of
    the persistent kernels changes with B; the arithmetic per patch must not), and patch 0 / 31 match the oracle."""
    case, sd, _, _ = model_case("DistgSSR", "full")
    rt = capi.DistgSSRRuntime(5, 4)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
    x = torch.from_numpy(synth_input((32, 1, 160, 160), seed=1)).cuda()
    y = rt.forward(x).clone()
    assert torch.isfinite(y).all()
    for i in range(32):
        assert torch.equal(rt.forward(x[i:i + 1]), y[i:i + 1]), i
    for i in (0, 31):
        ref = O.distgssr_forward(x[i:i + 1].cpu().numpy(), sd, 5, 4, dtype=np.float32)
        assert np.abs(y[i:i + 1].cpu().numpy() - ref).max() < 1e-4
