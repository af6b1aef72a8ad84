"""N2 / N3.  CPU part: the oracle's SSIM restatement against scipy's gaussian_filter (the routine skimage calls;
skimage itself is not installed, so the reference's cal_metrics cannot be run here: SSIM parity is pinned on scipy only).
GPU part: device metrics and view masking against the oracle / the reference's slice-assignment semantics."""
import random
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import lfsr_oracle as O


def test_oracle_gauss_matches_scipy():
    nd = pytest.importorskip("scipy.ndimage")
    img = np.random.default_rng(0).random((37, 29))
    assert np.abs(O._gauss11(img) - nd.gaussian_filter(img, sigma=1.5, truncate=3.5, mode="reflect")).max() < 1e-14


def test_oracle_ssim_identity_and_psnr():
    x = np.random.default_rng(1).random((1, 1, 2 * 16, 2 * 20))
    ps, ss = O.view_psnr_ssim(x, x, 2)
    assert np.isinf(ps).all() and np.allclose(ss, 1.0)
    y = x + 0.1
    ps, _ = O.view_psnr_ssim(x, y, 2)
    assert np.allclose(ps, 20.0)


@pytest.mark.gpu
def test_device_metrics_vs_oracle():
    from lfsr_amd import capi
    from lfsr_amd.utils.metrics import cal_metrics, view_metrics
    rng = np.random.default_rng(2)
    for (B, A, H, W) in [(2, 5, 128, 128), (1, 3, 24, 40), (1, 2, 11, 11)]:
        lab = rng.random((B, 1, A * H, A * W)).astype(np.float32)
        out = np.clip(lab + 0.05 * rng.standard_normal(lab.shape).astype(np.float32), 0, 1)
        ps, ss = view_metrics(torch.from_numpy(lab).cuda(), torch.from_numpy(out).cuda(), A)
        rps, rss = O.view_psnr_ssim(lab, out, A)
        assert np.abs(ps.cpu().numpy() - rps).max() < 1e-9
        assert np.abs(ss.cpu().numpy() - rss).max() < 1e-10
        pm, sm = cal_metrics(Namespace(angRes_in=A, task="SR"), torch.from_numpy(lab).cuda(), torch.from_numpy(out).cuda())
        assert abs(pm - rps.mean()) < 1e-9 and abs(sm - rss.mean()) < 1e-10
    with pytest.raises(capi.LfsrError):
        view_metrics(torch.zeros(1, 1, 10, 10, device="cuda"), torch.zeros(1, 1, 10, 10, device="cuda"), 2)   # 5x5 views < 11x11 window


@pytest.mark.gpu
def test_device_masking_matches_reference_semantics():
    from lfsr_amd.utils.masked_pretraining import MaskedAngularPretraining, ProgressiveMasking
    A, h, w = 5, 6, 8
    x = torch.rand(3, 1, A * h, A * w, device="cuda")
    for strategy in ("random", "grid", "corners", "center"):
        m = MaskedAngularPretraining(angRes=A, mask_ratio=0.3, mask_strategy=strategy).train()
        random.seed(4)
        applied = 0
        for _ in range(8):
            y, info = m(x)
            if not info["masked"]:
                assert y is x
                continue
            applied += 1
            ref = x.clone()
            for (i, j) in info["mask_indices"]:            # masked_pretraining.py:112-120
                ref[:, :, i * h:(i + 1) * h, j * w:(j + 1) * w] = 0
            assert torch.equal(y, ref)
            assert (A // 2, A // 2) not in info["mask_indices"]
            assert len(info["mask_indices"]) == min(max(1, int(25 * 0.3)), 4 if strategy == "corners" else 24)
        assert applied > 0
    m.eval()
    assert m(x)[0] is x                                   # train-time only
    pm = ProgressiveMasking(angRes=A, start_ratio=0.1, end_ratio=0.3, warmup_epochs=10)
    pm.set_epoch(5)
    assert pm.masker.num_masked == max(1, int(25 * 0.2))


# ---- N4: YCbCr -> RGB uint8 views + BMP files (train.py:329-341) -------------------------------------------------------------
def _read_bmp(path):
    import struct
    b = open(path, "rb").read()
    assert b[:2] == b"BM"
    off = struct.unpack("<I", b[10:14])[0]
    w, h, planes, bpp = struct.unpack("<iiHH", b[18:30])
    assert planes == 1 and bpp == 24 and h > 0
    row = (w * 3 + 3) // 4 * 4
    body = np.frombuffer(b, dtype=np.uint8, count=row * h, offset=off).reshape(h, row)[:, :w * 3].reshape(h, w, 3)
    return body[::-1, :, ::-1]   # bottom-up BGR -> top-down RGB


def test_oracle_ycbcr_roundtrip_and_bmp_writer(tmp_path):
    from lfsr_amd.utils.utils import rgb2ycbcr, write_bmp, save_views_bmp
    rng = np.random.default_rng(7)
    rgb = rng.random((9, 13, 3))
    ycc = rgb2ycbcr(rgb)
    assert np.abs(O.ycbcr2rgb(ycc) - rgb).max() < 1e-12          # the two reference formulas are inverses
    img = (rng.random((7, 5, 3)) * 255).astype(np.uint8)         # odd width: row padding
    write_bmp(str(tmp_path / "a.bmp"), img)
    assert np.array_equal(_read_bmp(str(tmp_path / "a.bmp")), img)
    views = (rng.random((2, 3, 4, 6, 3)) * 255).astype(np.uint8)
    save_views_bmp(tmp_path / "scene", views)
    assert np.array_equal(_read_bmp(str(tmp_path / "scene" / "View_1_2.bmp")), views[1, 2])


@pytest.mark.gpu
def test_device_ycbcr2rgb_views_vs_oracle():
    """bit-exact uint8 against the reference's float64 numpy arithmetic, incl. values on both sides of the clip"""
    from lfsr_amd.utils.utils import ycbcr2rgb_views
    rng = np.random.default_rng(8)
    for (A, h, w) in [(5, 32, 32), (3, 7, 10), (2, 128, 96)]:
        y = (rng.random((A * h, A * w)) * 1.2 - 0.1).astype(np.float32)
        cc = (rng.random((2, A * h, A * w)) * 1.2 - 0.1).astype(np.float32)
        got = ycbcr2rgb_views(torch.from_numpy(y)[None, None].cuda(), torch.from_numpy(cc)[None].cuda(), A).cpu().numpy()
        ref = O.sr_views_rgb_u8(y.astype(np.float64), cc.astype(np.float64), A)
        assert got.shape == (A, A, h, w, 3) and np.array_equal(got, ref)
