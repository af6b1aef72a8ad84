"""GPU: a1-a7 through the C ABI, bit-exact against the numpy oracle and the golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from oracle import lfsr_oracle as O
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
IDX = np.load(os.path.join(GOLDEN, "index_ops.npz"))
META = json.load(open(os.path.join(GOLDEN, "index_ops.json")))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def same(t, ref):
    return np.array_equal(t.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int32])
@pytest.mark.parametrize("shape", [(2, 3, 5, 4, 6), (1, 2, 3, 7, 5), (2, 64, 5, 32, 32), (1, 1, 1, 3, 3), (3, 1, 7, 2, 9)])
def test_sai_macpi(shape, dtype):
    B, C, A, h, w = shape
    rng = np.random.default_rng(0)
    x = (rng.random((B, C, A * h, A * w)) * 1000).astype(dtype)
    assert same(capi.sai2macpi(dev(x), A), O.sai2macpi(x, A))
    assert same(capi.macpi2sai(dev(x), A), O.macpi2sai(x, A))
    assert same(capi.macpi2sai(capi.sai2macpi(dev(x), A), A), x)


@pytest.mark.parametrize("tag", ["s2m_a", "s2m_b"])
def test_sai_macpi_golden(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["A"] * m["h"] * m["A"] * m["w"], dtype=np.int32).reshape(m["B"], m["C"], m["A"] * m["h"], m["A"] * m["w"])
    assert same(capi.sai2macpi(dev(x), m["A"]), IDX[tag + "_sai2macpi"])
    assert same(capi.macpi2sai(dev(x), m["A"]), IDX[tag + "_macpi2sai"])


def test_empty_batch():
    x = torch.zeros((0, 3, 10, 10), device="cuda")
    assert capi.sai2macpi(x, 5).shape == (0, 3, 10, 10)
    assert capi.pixel_shuffle2d(torch.zeros((0, 8, 2, 2), device="cuda"), 2).shape == (0, 2, 4, 4)


@pytest.mark.parametrize("shape", [(2, 3, 5, 3, 4), (1, 2, 4, 5, 3), (1, 4, 2, 2, 2), (2, 64, 4, 40, 40), (1, 16, 5, 32, 32)])
def test_pixel_shuffle2d(shape):
    B, C, r, h, w = shape
    x = np.random.default_rng(1).random((B, C * r * r, h, w)).astype(np.float32)
    assert same(capi.pixel_shuffle2d(dev(x), r), O.pixel_shuffle(x, r))
    assert same(capi.pixel_shuffle2d(dev(x), r), torch.nn.PixelShuffle(r)(torch.from_numpy(x)).numpy())


@pytest.mark.parametrize("tag", ["ps_a", "ps_b", "ps_c"])
def test_pixel_shuffle2d_golden(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["r"] ** 2 * m["h"] * m["w"], dtype=np.int32).reshape(m["B"], m["C"] * m["r"] ** 2, m["h"], m["w"])
    assert same(capi.pixel_shuffle2d(dev(x), m["r"]), IDX[tag])


@pytest.mark.parametrize("shape", [(2, 3, 5, 4, 3), (1, 2, 3, 2, 5), (2, 32, 5, 160, 32)])
def test_pixel_shuffle1d(shape):
    B, C, f, h, w = shape
    x = np.random.default_rng(2).random((B, C * f, h, w)).astype(np.float32)
    assert same(capi.pixel_shuffle1d(dev(x), f), O.pixel_shuffle1d(x, f))


@pytest.mark.parametrize("tag", ["ps1d_a", "ps1d_b"])
def test_pixel_shuffle1d_golden(tag):
    m = META[tag]
    x = np.arange(m["B"] * m["C"] * m["f"] * m["h"] * m["w"], dtype=np.int32).reshape(m["B"], m["C"] * m["f"], m["h"], m["w"])
    assert same(capi.pixel_shuffle1d(dev(x), m["f"]), IDX[tag])


def test_image_extend():
    m = META["imext"]
    x = np.arange(np.prod(m["shape"]), dtype=np.int32).reshape(m["shape"])
    assert same(capi.image_extend(dev(x), m["bdr"]), IDX["imext"])
    y = np.random.default_rng(3).random((25, 1, 40, 33)).astype(np.float32)
    assert same(capi.image_extend(dev(y), [8, 23, 8, 23]), O.image_extend(y, [8, 23, 8, 23]))


@pytest.mark.parametrize("key", sorted(META["lfdivide"].keys()))
def test_lfdivide_integrate(key):
    import hashlib
    m = META["lfdivide"][key]
    A, h0, w0, P, S = m["A"], m["h0"], m["w0"], m["P"], m["S"]
    x = np.arange(A * h0 * A * w0, dtype=np.int32).reshape(A * h0, A * w0)
    sub = capi.lf_divide(dev(x), A, P, S)
    assert tuple(sub.shape[:2]) == (m["numU"], m["numV"])
    sub_np = sub.cpu().numpy()
    assert hashlib.sha256(sub_np.tobytes()).hexdigest() == m["divide_sha"]       # reference output checksum
    assert np.array_equal(sub_np, O.lf_divide(x, A, P, S))
    back = capi.lf_integrate(sub, A, P, S, h0, w0)                              # size-independent property: round trip
    assert same(back.permute(0, 2, 1, 3).reshape(A * h0, A * w0), x)
    s = 4
    big = (np.arange(sub_np.size * s * s, dtype=np.int64) % 16777213).astype(np.int32).reshape(
        sub_np.shape[0], sub_np.shape[1], sub_np.shape[2] * s, sub_np.shape[3] * s)
    integ = capi.lf_integrate(dev(big), A, P * s, S * s, h0 * s, w0 * s).cpu().numpy()
    assert hashlib.sha256(integ.tobytes()).hexdigest() == m["integrate_s4_sha"]


def test_lfdivide_float_and_6d_integrate():
    A, h0, w0 = 5, 45, 52
    x = np.random.default_rng(4).random((A * h0, A * w0)).astype(np.float32)
    sub = capi.lf_divide(dev(x), A, 32, 16)
    assert same(sub, O.lf_divide(x, A, 32, 16))
    n1, n2 = sub.shape[:2]
    six = sub.reshape(n1, n2, A, 32, A, 32).permute(0, 1, 2, 4, 3, 5).contiguous()
    assert same(capi.lf_integrate(six, A, 32, 16, h0, w0), O.lf_integrate(sub.cpu().numpy(), A, 32, 16, h0, w0))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", [(2, 64, 5, 8, 8), (1, 144, 3, 6, 8), (1, 16, 5, 32, 32), (1, 1, 3, 5, 7)])
def test_nchw_vcl_roundtrip(shape, layout):
    B, C, A, h, w = shape
    x = np.random.default_rng(5).random((B, C, A * h, A * w)).astype(np.float32)
    v = capi.nchw_to_vcl(dev(x), A, layout)
    # VCL definition: [b][u*A+v][y][x][c]
    xs = x.reshape(B, C, A, h, A, w) if layout == 0 else x.reshape(B, C, h, A, w, A).transpose(0, 1, 3, 2, 5, 4)
    ref = xs.transpose(0, 2, 4, 3, 5, 1).reshape(-1, C)
    assert same(v, ref)
    assert same(capi.vcl_to_nchw(v, B, C, A, h, w, layout), x)
