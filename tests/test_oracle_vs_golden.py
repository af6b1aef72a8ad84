"""CPU: the numpy oracle against golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  Integer ops bit-exact; float ops to fp32 round-off."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import lfsr_oracle as O
from tests.helpers import GOLDEN, model_case, psnr

IDX = np.load(os.path.join(GOLDEN, "index_ops.npz"))
META = json.load(open(os.path.join(GOLDEN, "index_ops.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("tag", ["s2m_a", "s2m_b"])
def test_sai_macpi(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["A"] * m["h"] * m["A"] * m["w"], dtype=np.int32).reshape(
        m["B"], m["C"], m["A"] * m["h"], m["A"] * m["w"])
    assert np.array_equal(O.sai2macpi(x, m["A"]), IDX[tag + "_sai2macpi"])
    assert np.array_equal(O.macpi2sai(x, m["A"]), IDX[tag + "_macpi2sai"])
    assert np.array_equal(O.macpi2sai(O.sai2macpi(x, m["A"]), m["A"]), x)


@pytest.mark.parametrize("tag", ["ps_a", "ps_b", "ps_c"])
def test_pixel_shuffle(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["r"] ** 2 * m["h"] * m["w"], dtype=np.int32).reshape(
        m["B"], m["C"] * m["r"] ** 2, m["h"], m["w"])
    assert np.array_equal(O.pixel_shuffle(x, m["r"]), IDX[tag])


@pytest.mark.parametrize("tag", ["ps1d_a", "ps1d_b"])
def test_pixel_shuffle1d(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["f"] * m["h"] * m["w"], dtype=np.int32).reshape(
        m["B"], m["C"] * m["f"], m["h"], m["w"])
    assert np.array_equal(O.pixel_shuffle1d(x, m["f"]), IDX[tag])


def test_image_extend():
    m = META["imext"]
    x = np.arange(np.prod(m["shape"]), dtype=np.int32).reshape(m["shape"])
    assert np.array_equal(O.image_extend(x, m["bdr"]), IDX["imext"])


@pytest.mark.parametrize("key", sorted(META["lfdivide"].keys()))
def test_lfdivide_integrate(key):
    m = META["lfdivide"][key]
    A, h0, w0, P, S = m["A"], m["h0"], m["w0"], m["P"], m["S"]
    x = np.arange(A * h0 * A * w0, dtype=np.int32).reshape(A * h0, A * w0)
    sub = O.lf_divide(x, A, P, S)
    assert sub.shape[:2] == (m["numU"], m["numV"])
    assert sha(sub) == m["divide_sha"]
    if "div_" + key in IDX.files:
        assert np.array_equal(sub, IDX["div_" + key])
    # round trip at scale 1 (the reference's own free property)
    back = O.lf_integrate(sub, A, P, S, h0, w0)
    assert np.array_equal(back.transpose(0, 2, 1, 3).reshape(A * h0, A * w0), x)
    # x4 integrate on a known ramp
    s = 4
    big = (np.arange(sub.size * s * s, dtype=np.int64) % 16777213).astype(np.int32).reshape(
        sub.shape[0], sub.shape[1], sub.shape[2] * s, sub.shape[3] * s)
    integ = O.lf_integrate(big, A, P * s, S * s, h0 * s, w0 * s)
    assert list(integ.shape) == m["integrate_s4_shape"]
    assert sha(integ) == m["integrate_s4_sha"]


DISTG_TAPS = {"init_conv_0": "init_conv", "b0_spa_0": "b0_spa", "b0_ang_0": "b0_ang", "b0_epi_last_0": "b0_epih",
              "b0_epi_last_1": "b0_epiv_t", "b0_out_0": "b0_out", "g0_out_0": "g0_out", "disentg_out_0": "disentg_out"}


@pytest.mark.parametrize("tag", ["a5h8s4", "a3h6w8s2"])
def test_distgssr_small(tag):
    case, sd, x, npz = model_case("DistgSSR", tag)
    taps = {}
    y = O.distgssr_forward(x, sd, case["A"], case["s"], taps=taps)
    g = npz[tag + "_out"]
    assert y.shape == g.shape
    assert np.abs(y - g).max() < 2e-5
    assert psnr(y, g) > 100.0
    if tag == "a5h8s4":
        for gk, ok in DISTG_TAPS.items():
            ref = npz[f"{tag}_{gk}"]
            assert np.abs(taps[ok] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), gk


def test_distgssr_fp32_mode():
    case, sd, x, npz = model_case("DistgSSR", "a3h6w8s2")
    y = O.distgssr_forward(x, sd, case["A"], case["s"], dtype=np.float32)
    assert y.dtype == np.float32
    assert np.abs(y - npz["a3h6w8s2_out"]).max() < 5e-5


def test_interp_against_torch():
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    x = np.random.default_rng(0).random((2, 1, 7, 9))
    for s in (2, 4):
        tb = F.interpolate(torch.from_numpy(x), scale_factor=s, mode="bilinear", align_corners=False).numpy()
        tc = F.interpolate(torch.from_numpy(x), scale_factor=s, mode="bicubic", align_corners=False).numpy()
        assert np.abs(O.interp_bilinear(x, s) - tb).max() < 1e-12
        assert np.abs(O.interp_bicubic(x, s) - tc).max() < 1e-12


@pytest.mark.parametrize("tag", ["a5h8s4", "a3h6w8s2"])
def test_torch_port_distgssr(tag):
    """the torch-CPU form of the oracle (bench.py's cpu_baseline) against the reference's output"""
    torch = pytest.importorskip("torch")
    from oracle import lfsr_torch_port as T
    case, sd, x, npz = model_case("DistgSSR", tag)
    y = T.distgssr_forward(torch.from_numpy(x), {k: torch.from_numpy(v) for k, v in sd.items()}, case["A"], case["s"]).numpy()
    assert np.abs(y - npz[tag + "_out"]).max() < 1e-5


@pytest.mark.parametrize("name,tag", [("EPIT", "a5h8s4"), ("EPIT", "a3h6w8s2"), ("LFT", "a5h8s4"), ("LFT", "a3h6w8s2"),
                                      ("LF_InterNet", "a5h8s2"), ("LF_InterNet", "a3h6w8s4")])
def test_torch_port_other_models(name, tag):
    """the torch-CPU forms timed as bench.py's cpu_baseline lines for configs 1, 3, 5 against the reference's outputs"""
    torch = pytest.importorskip("torch")
    from oracle import lfsr_torch_port as T
    fn = {"EPIT": T.epit_forward, "LFT": T.lft_forward, "LF_InterNet": T.internet_forward}[name]
    case, sd, x, npz = model_case(name, tag)
    y = fn(torch.from_numpy(x), {k: torch.from_numpy(v) for k, v in sd.items()}, case["A"], case["s"]).numpy()
    assert np.abs(y - npz[tag + "_out"]).max() < 1e-5


@pytest.mark.parametrize("tag", ["a5h8s4", "a3h6w8s2"])
def test_epit_small(tag):
    case, sd, x, npz = model_case("EPIT", tag)
    y = O.epit_forward(x, sd, case["A"], case["s"])
    g = npz[tag + "_out"]
    assert y.shape == g.shape
    assert np.abs(y - g).max() < 5e-5
    assert psnr(y, g) > 95.0


@pytest.mark.parametrize("tag", ["a5h8s4", "a3h6w8s2"])
def test_lft_small(tag):
    case, sd, x, npz = model_case("LFT", tag)
    y = O.lft_forward(x, sd, case["A"], case["s"])
    g = npz[tag + "_out"]
    assert y.shape == g.shape
    assert np.abs(y - g).max() < 5e-5
    assert psnr(y, g) > 95.0


@pytest.mark.parametrize("tag", ["a5h8s2", "a3h6w8s4"])
def test_internet_small(tag):
    case, sd, x, npz = model_case("LF_InterNet", tag)
    y = O.internet_forward(x, sd, case["A"], case["s"])
    g = npz[tag + "_out"]
    assert y.shape == g.shape
    assert np.abs(y - g).max() < 5e-5 * max(1.0, np.abs(g).max())


def test_winograd_restatement_equals_direct_conv():
    """the F(2x2,3x3) algebra the HIP conv kernel implements (oracle.conv3x3_winograd) is the same correlation as conv2d"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 8, 6, 8))
    w = rng.standard_normal((4, 8, 3, 3))
    ref = O.conv2d(x, w, padding=(1, 1))
    assert np.abs(O.conv3x3_winograd(x, w) - ref).max() < 1e-12
    # fp32 round-off of the Winograd form stays at fp32 epsilon scale on unit-variance data
    y32 = O.conv3x3_winograd(x.astype(np.float32), w.astype(np.float32))
    assert np.abs(y32 - ref).max() < 5e-5
    # packed layout: entry (j, nt, p, half, n32, e) holds U[p][n = 32 nt + n32][k = 8 j + 4 half + e]
    w64 = rng.standard_normal((64, 64, 3, 3)).astype(np.float32)
    pk = O.winograd_pack(w64).reshape(8, 2, 16, 2, 32, 4)
    U = O.winograd_weights(w64.astype(np.float64)).reshape(16, 64, 64)
    for (j, nt, p, hf, n32, e) in [(0, 0, 0, 0, 0, 0), (7, 1, 15, 1, 31, 3), (3, 0, 6, 1, 17, 2), (5, 1, 9, 0, 4, 1)]:
        assert pk[j, nt, p, hf, n32, e] == np.float32(U[p, 32 * nt + n32, 8 * j + 4 * hf + e])


def test_winograd4_restatement_equals_direct_conv():
    """the F(4x4,3x3) algebra of the default HIP conv kernel (oracle.conv3x3_winograd4) is the same correlation as conv2d"""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 8, 8, 12))
    w = rng.standard_normal((4, 8, 3, 3))
    ref = O.conv2d(x, w, padding=(1, 1))
    assert np.abs(O.conv3x3_winograd4(x, w) - ref).max() < 1e-11
    y32 = O.conv3x3_winograd4(x.astype(np.float32), w.astype(np.float32))
    assert np.abs(y32 - ref).max() < 2e-4          # fp32 round-off of F(4x4,3x3) on unit-variance data (|y| ~ 8)
    # packed layout: entry (s, ns, q, lane, e) holds U of place p = 4 q + e for [n = 16 ns + lane % 16][k = 4 s + lane / 16]; place p = 18 (nu // 3) + 3 xi + nu % 3
    # (nu-half major: the 18 positions a half tile's wave owns are contiguous)
    w64 = rng.standard_normal((64, 64, 3, 3)).astype(np.float32)
    pk = O.winograd4_pack(w64).reshape(16, 4, 9, 64, 4)
    U = O.winograd4_weights(w64.astype(np.float64))          # [xi][nu][n][k]
    for (s_, ns, q, ln, e) in [(0, 0, 0, 0, 0), (15, 3, 8, 63, 3), (3, 1, 5, 37, 2), (9, 2, 7, 16, 1), (2, 0, 4, 5, 1), (2, 0, 4, 5, 2)]:
        pl = 4 * q + e
        hf, xi, n3 = pl // 18, (pl % 18) // 3, pl % 3
        assert pk[s_, ns, q, ln, e] == np.float32(U[xi, 3 * hf + n3, 16 * ns + ln % 16, 4 * s_ + ln // 16])


def test_winograd4_roundoff_through_the_whole_network():
    """fp32 F(4x4,3x3) in every 3x3 64->64 conv of the DistgSSR forward (53 of them) stays at fp32 round-off of the fp64 direct form --
    the measurement that cleared the HIP kernel's algorithm before it was written"""
    import torch
    import torch.nn.functional as F
    import oracle.lfsr_torch_port as P
    from tests.helpers import model_case, psnr
    case, sd, x, gold = model_case("DistgSSR", "a5h8s4")
    A, s_ = case["A"], case["s"]
    BT, AT = torch.from_numpy(O.WINO4_BT).float(), torch.from_numpy(O.WINO4_AT).float()

    def wino4(xv, w):   # xv (N, C, h, w) view images, w (O, C, 3, 3): U in fp64 rounded once, everything else fp32
        U = torch.from_numpy(O.winograd4_weights(w.double().numpy())).float()
        N, C, h, wd = xv.shape
        pat = F.pad(xv, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)          # N, C, th, tw, 6, 6
        V = torch.einsum("ai,ncyxij,bj->abncyx", BT, pat, BT)
        M = torch.einsum("aboc,abncyx->abnoyx", U, V)
        return torch.einsum("ia,abnoyx,jb->noyixj", AT, M, AT).reshape(N, U.shape[2], h, wd)

    orig = F.conv2d
    use = {"on": False}

    def patched(t, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if use["on"] and w.shape[1] == 64 and tuple(w.shape[2:]) == (3, 3):
            v = P.macpi2sai(t, A)
            B_, C, HH, WW = v.shape
            h, wd = HH // A, WW // A
            vv = v.reshape(B_, C, A, h, A, wd).permute(0, 2, 4, 1, 3, 5).reshape(B_ * A * A, C, h, wd)
            y = wino4(vv, w).reshape(B_, A, A, -1, h, wd).permute(0, 3, 1, 4, 2, 5).reshape(B_, -1, HH, WW)
            return P.sai2macpi(y, A)
        return orig(t, w, b, stride, padding, dilation, groups)
    P.F.conv2d = patched
    try:
        sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
        sd32 = {k: torch.from_numpy(v) for k, v in sd.items()}
        ref = P.distgssr_forward(torch.from_numpy(x).double(), sd64, A, s_).numpy()
        use["on"] = True
        y = P.distgssr_forward(torch.from_numpy(x), sd32, A, s_).double().numpy()
    finally:
        P.F.conv2d = orig
    assert np.abs(y - ref).max() < 1e-5 and psnr(y, ref) > 120.0


def test_split_bf16_products():
    """the arithmetic of the b3 kernels (rowgemm_b3.hip, epi_b3.hip, ffn_b3.hip, ...) restated in numpy: fp32 operands as three bf16 terms (input: truncation, exact; weights: round to
    nearest), six of the nine cross products -> closer to the fp64 dot product than an fp32 dot product is, and the split itself is exact"""
    rng = np.random.default_rng(0)
    def trunc(x): return (x.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    def rne(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)
    def split3(x, f):
        a = f(x); r = (x - a).astype(np.float32); b = f(r); r2 = (r - b).astype(np.float32); return a, b, f(r2)
    x = (rng.standard_normal((500, 64)) * 3).astype(np.float32); w = (rng.uniform(-1, 1, (64, 64)) / 24).astype(np.float32)
    x0, x1, x2 = split3(x, trunc); w0, w1, w2 = split3(w, rne)
    assert np.array_equal(x0.astype(np.float64) + x1 + x2, x.astype(np.float64))
    assert np.array_equal(w0.astype(np.float64) + w1 + w2, w.astype(np.float64))
    ref = x.astype(np.float64) @ w.astype(np.float64)
    six = sum(a.astype(np.float64) @ b.astype(np.float64) for a, b in ((x0, w0), (x1, w1), (x1, w0), (x0, w1), (x0, w2), (x2, w0)))
    e6, e32 = np.abs(six - ref).max() / np.abs(ref).max(), np.abs((x @ w) - ref).max() / np.abs(ref).max()
    assert e6 < 1e-7 and e6 < e32
