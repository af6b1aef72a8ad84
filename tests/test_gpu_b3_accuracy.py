"""GPU: the kernels that carry fp32 operands as three exact bf16 terms on the bf16 MFMA pipe (rowgemm_b3.hip, ffn_b3.hip, epi_b3.hip) against fp64,
with the fp32-MFMA kernel of the same operator as the yardstick (the three-term form may stand in for fp32 as long as its error against
fp64 is no larger than the fp32 kernel's).  Random shapes (ragged M, both K, residual / none), unit-variance data with a per-row offset; every
case also holds the 1e-4 absolute gate of the operator tests.  The reference computes these layers with stock fp32 torch ops
(EPIT.py:110-128, LFT.py:188-246, DistgSSR.py:91-97)."""
import os

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from oracle import lfsr_oracle as O

pytestmark = pytest.mark.gpu
F32 = {"LFSR_ROWGEMM": "f32", "LFSR_FFN": "f32", "LFSR_EPI": "wino"}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def both(fn, monkeypatch):
    """fn() under the default (three-term bf16) selection and under the fp32-MFMA selection"""
    for k in F32:
        monkeypatch.delenv(k, raising=False)
    a = fn()
    for k, v in F32.items():
        monkeypatch.setenv(k, v)
    b = fn()
    for k in F32:
        monkeypatch.delenv(k, raising=False)
    return a, b


def test_linear_and_ffn_sweep(monkeypatch):
    lib = capi.load()
    rng = np.random.default_rng(2026)
    worst = {}
    for it in range(10):
        K = int(rng.choice([64, 128])); M = int(rng.integers(2048, 30000)); N = int(rng.choice([64, 128, 192, 256, 384]))
        x = (rng.standard_normal((M, K)) + rng.standard_normal((M, 1))).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
        r = rng.standard_normal((M, N)).astype(np.float32); use_r = bool(rng.integers(0, 2))
        xd, rd = dev(x), dev(r)
        wp = capi.pack_conv_weight(dev(w.reshape(N, K, 1, 1)))

        def lin():
            y = torch.full((M, N), float("nan"), device="cuda")
            capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), None, capi.dev_ptr(rd) if use_r else None, N, 0, capi.dev_ptr(y), N, 0, M, N, 1.0,
                                           capi.stream_ptr()), "linear")
            return y.cpu().numpy().astype(np.float64)
        ref = x.astype(np.float64) @ w.astype(np.float64).T + (r if use_r else 0.0)
        yb, yf = both(lin, monkeypatch)
        eb, ef = np.abs(yb - ref), np.abs(yf - ref)
        assert eb.max() < 1e-4
        worst.setdefault("linear", []).append((eb.mean(), ef.mean(), eb.max(), ef.max()))
        # fused LayerNorm + feed-forward
        H = 2 * K
        g = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32); b = (0.2 * rng.standard_normal(K)).astype(np.float32)
        w1 = (rng.standard_normal((H, K)) * 0.1).astype(np.float32); w2 = (rng.standard_normal((K, H)) * 0.1).astype(np.float32)
        w1p = capi.pack_conv_weight(dev(w1.reshape(H, K, 1, 1))); w2p = capi.pack_conv_weight(dev(w2.reshape(K, H, 1, 1)))
        gd, bd = dev(g), dev(b)

        def ffn():
            yf_ = torch.full((M, K), float("nan"), device="cuda")
            capi.check(lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K, 0, capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(xd), K, 0,
                                           capi.dev_ptr(yf_), K, 0, M, K, H, K, 0.0, capi.stream_ptr()), "ffn_ln")
            return yf_.cpu().numpy().astype(np.float64)
        x64 = x.astype(np.float64)
        xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
        reff = np.maximum(xn @ w1.astype(np.float64).T, 0.0) @ w2.astype(np.float64).T + x64
        yb, yf = both(ffn, monkeypatch)
        eb, ef = np.abs(yb - reff), np.abs(yf - reff)
        assert eb.max() < 1e-4
        worst.setdefault("ffn_ln", []).append((eb.mean(), ef.mean(), eb.max(), ef.max()))
    for k, rows in worst.items():
        a = np.array(rows)
        print(f"{k}: mean |err| three-term {a[:, 0].mean():.2e} vs fp32 MFMA {a[:, 1].mean():.2e}; max {a[:, 2].max():.2e} vs {a[:, 3].max():.2e}")
        # yardstick: over the sweep the three-term form's mean error is not above the fp32-MFMA kernel's (10 % slack for the LayerNorm's shared fp32 part)
        assert a[:, 0].mean() <= 1.1 * a[:, 1].mean(), k
        assert a[:, 2].max() <= 1.5 * a[:, 3].max(), k


def test_fuse0_three_term_vs_fp32_kernel(monkeypatch):
    """DistgSSR's fuse.0 (1x1 144 -> 64 + LeakyReLU, DistgSSR.py:99) on k_rowgemm_b3 with 144 of 160 operand columns valid, read out of a wider buffer whose
    neighbouring rows hold NaN-free but LARGE values (the masked columns must not leak), against fp64 and the fp32-MFMA row-GEMM"""
    rng = np.random.default_rng(144)
    M = 5003
    xw = (rng.standard_normal((M, 160)) * 1e3).astype(np.float32)          # row stride 160: columns 8 .. 151 are the operand, the rest is foreign data
    xw[:, 8:152] = rng.standard_normal((M, 144)).astype(np.float32) + rng.standard_normal((M, 1)).astype(np.float32)
    w = (rng.standard_normal((64, 144)) * 0.1).astype(np.float32)
    lib = capi.load()
    xd = dev(xw)
    wp = capi.pack_conv_weight(dev(w.reshape(64, 144, 1, 1)))

    def run():
        y = torch.full((M, 64), float("nan"), device="cuda")
        capi.check(lib.lfsr_pointwise_fwd(capi.dev_ptr(xd), 160, 8, 144, capi.dev_ptr(wp), None, capi.dev_ptr(y), 64, 0, M, 64, 0.1, capi.stream_ptr()), "pointwise")
        return y.cpu().numpy().astype(np.float64)
    pre = xw[:, 8:152].astype(np.float64) @ w.astype(np.float64).T
    ref = np.where(pre >= 0, pre, 0.1 * pre)
    yb, yf = both(run, monkeypatch)
    eb, ef = np.abs(yb - ref), np.abs(yf - ref)
    print(f"fuse.0: three-term mean {eb.mean():.2e} max {eb.max():.2e} | fp32 MFMA mean {ef.mean():.2e} max {ef.max():.2e}")
    assert eb.max() < 1e-4 and eb.mean() <= 1.1 * ef.mean() and eb.max() <= 1.5 * ef.max()
    assert not np.array_equal(yb, yf)


@pytest.mark.parametrize("B,h,w", [(2, 32, 32), (1, 17, 32), (1, 32, 9)])
def test_epi_branch_three_term_vs_fp32_kernel(B, h, w, monkeypatch):
    """both EPI passes (DistgSSR.py:91-97,108) at angRes 5, unit-variance input: error against fp64 of epi_b3.hip and of the fp32-MFMA Winograd kernel"""
    A = 5
    rng = np.random.default_rng(B * 100 + h + w)
    x = rng.standard_normal((B, 64, A * h, A * w)).astype(np.float32)
    w1 = (rng.standard_normal((32, 64, 1, A * A)) * 0.03).astype(np.float32)
    w2 = (rng.standard_normal((A * 32, 32, 1, 1)) * 0.15).astype(np.float32)

    def epi(t):
        e = O.leaky_relu(O.conv2d(t, w1.astype(np.float64), stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        return O.pixel_shuffle1d(O.leaky_relu(O.conv2d(e, w2.astype(np.float64)), 0.1), A)
    x64 = x.astype(np.float64)
    refh = epi(x64)
    refv = epi(np.ascontiguousarray(x64.transpose(0, 1, 3, 2))).transpose(0, 1, 3, 2)
    xv = capi.nchw_to_vcl(dev(x), A, 1)
    w1p, w2p = capi.pack_conv_weight(dev(w1)), capi.pack_conv_weight(dev(w2))

    def run():
        out = torch.full((B * A * A * h * w, 64), float("nan"), device="cuda")
        capi.epiconv_hv(xv, w1p, w2p, B, A, h, w, 0.1, out, 0, 32)
        return (capi.vcl_to_nchw(out, B, 32, A, h, w, 1, 0).cpu().numpy().astype(np.float64), capi.vcl_to_nchw(out, B, 32, A, h, w, 1, 32).cpu().numpy().astype(np.float64))
    (bh, bv), (fh, fv) = both(run, monkeypatch)
    eb = np.concatenate([np.abs(bh - refh).ravel(), np.abs(bv - refv).ravel()])
    ef = np.concatenate([np.abs(fh - refh).ravel(), np.abs(fv - refv).ravel()])
    print(f"EPI branch {B}x{h}x{w}: three-term mean {eb.mean():.2e} max {eb.max():.2e} | fp32 MFMA (F(2,5)) mean {ef.mean():.2e} max {ef.max():.2e}")
    assert eb.max() < 1e-4
    assert eb.mean() <= 1.1 * ef.mean() and eb.max() <= 1.5 * ef.max()
    assert not np.array_equal(bh, fh)            # the two selections really ran different kernels


def test_epi_branch_batch8_equals_smaller_batches():
    """B = 8 at angRes 5, 32x32 (320 groups of 8 EPI lines on 256 CUs: the persistent kernel's leftover groups are cut into 2-line sub-groups) against the same items run
    in batches of 2 and 3 (whole groups only): bit-equal -- the unit decomposition changes no line's arithmetic"""
    A, h, w, B = 5, 32, 32, 8
    g = torch.Generator().manual_seed(88)
    x = torch.randn(B, 64, A * h, A * w, generator=g)
    w1p = capi.pack_conv_weight((torch.randn(32, 64, 1, A * A, generator=g) * 0.03).cuda())
    w2p = capi.pack_conv_weight((torch.randn(A * 32, 32, 1, 1, generator=g) * 0.15).cuda())

    def run(xb):
        Bb = xb.shape[0]
        xv = capi.nchw_to_vcl(xb.cuda().contiguous(), A, 1)
        out = torch.full((xv.shape[0], 64), float("nan"), device="cuda")
        capi.epiconv_hv(xv, w1p, w2p, Bb, A, h, w, 0.1, out, 0, 32)
        return out.reshape(Bb, -1)
    full = run(x)
    assert torch.isfinite(full).all()
    parts = torch.cat([run(x[0:2]), run(x[2:5]), run(x[5:8])], 0)
    assert torch.equal(full, parts)


def test_rows_past_m_are_neither_written_nor_needed(monkeypatch):
    """The row-streaming kernels drop rows >= M of the last tile through their buffer descriptors (the bounds check covers VGPR / immediate offsets only: the tile base
    must not sit in the SGPR offset).  Output buffers carry a guard band of sentinel rows behind row M that must survive; the rows of x behind M hold NaN (a kernel that
    folded them into valid rows would show it); ragged M around every tile size."""
    lib = capi.load()
    rng = np.random.default_rng(7)
    K, N, G = 128, 384, 300
    w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    wp = capi.pack_conv_weight(dev(w.reshape(N, K, 1, 1)))
    g = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32); b = (0.2 * rng.standard_normal(K)).astype(np.float32)
    gd, bd = dev(g), dev(b)
    w1 = (rng.standard_normal((2 * K, K)) * 0.1).astype(np.float32); w2 = (rng.standard_normal((K, 2 * K)) * 0.1).astype(np.float32)
    w1p = capi.pack_conv_weight(dev(w1.reshape(2 * K, K, 1, 1))); w2p = capi.pack_conv_weight(dev(w2.reshape(K, 2 * K, 1, 1)))
    for M in (1, 15, 16, 17, 63, 65, 127, 129, 255, 257, 1000, 4099):
        x = rng.standard_normal((M + G, K)).astype(np.float32)
        x[M:] = np.nan
        xd = dev(x)
        x64 = x[:M].astype(np.float64)
        xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
        # linear (with a residual), LayerNorm + in-projection (two outputs), LayerNorm + feed-forward
        y = torch.full((M + G, 128), 7.5, device="cuda")
        capi.check(lib.lfsr_linear_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), None, capi.dev_ptr(xd), K, 0, capi.dev_ptr(y), 128, 0, M, 128, 1.0, capi.stream_ptr()), "linear")
        yh = y.cpu().numpy()
        assert (yh[M:] == 7.5).all(), M
        assert np.abs(yh[:M] - (x64 @ w[:128].astype(np.float64).T + x64)).max() < 1e-4, M
        qk = torch.full((M + G, 256), 7.5, device="cuda"); v = torch.full((M + G, 128), 7.5, device="cuda")
        capi.check(lib.lfsr_linear_ln_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, 256, None, 0, 0, 0,
                                          capi.dev_ptr(qk), 256, 0, capi.dev_ptr(v), 128, 0, 256, M, N, capi.stream_ptr()), "linear_ln")
        qh, vh = qk.cpu().numpy(), v.cpu().numpy()
        assert (qh[M:] == 7.5).all() and (vh[M:] == 7.5).all(), M
        assert np.abs(qh[:M] - xn @ w[:256].astype(np.float64).T).max() < 1e-4 and np.abs(vh[:M] - x64 @ w[256:].astype(np.float64).T).max() < 1e-4, M
        yf = torch.full((M + G, K), 7.5, device="cuda")
        capi.check(lib.lfsr_ffn_ln_fwd(capi.dev_ptr(xd), K, 0, capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, capi.dev_ptr(w1p), capi.dev_ptr(w2p), capi.dev_ptr(xd), K, 0,
                                       capi.dev_ptr(yf), K, 0, M, K, 2 * K, K, 0.0, capi.stream_ptr()), "ffn_ln")
        fh = yf.cpu().numpy()
        assert (fh[M:] == 7.5).all(), M
        assert np.abs(fh[:M] - (np.maximum(xn @ w1.astype(np.float64).T, 0.0) @ w2.astype(np.float64).T + x64)).max() < 1e-4, M


@pytest.mark.parametrize("K,ln_cols,split,with_pe", [(128, 256, 256, True), (128, 256, 256, False), (128, 128, 256, True), (128, 384, 128, False), (128, 0, 0, False),
                                                     (128, 320, 192, True), (128, 64, 0, False), (64, 128, 128, True), (64, 128, 128, False), (64, 64, 0, True), (64, 192, 64, False)])
def test_lnlin_weights_in_registers_form(K, ln_cols, split, with_pe, monkeypatch):
    """lnlin_b3.hip (K = 128, N = 384: the weight planes in registers, rows normed / split once through LDS) against fp64 for every mix of LayerNorm'd and raw column tiles
    inside a wave (ln_cols = 64 / 256: one LayerNorm'd tile of a wave's three, 128 / 320: two; 0 / 384: none / all), one or two outputs, ragged M around the 32-row stage, with and
    without a positional encoding whose row index steps with the block's stages; where the panel form (LFSR_LNLIN=0) covers the shape the two agree to rounding."""
    lib = capi.load()
    rng = np.random.default_rng(384 + ln_cols + K)
    N = 3 * K                                  # (128, 384): eight waves of three column tiles; (64, 192): four (LFT's angular transformer)
    w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    wp = capi.pack_conv_weight(dev(w.reshape(N, K, 1, 1)))
    g = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32); b = (0.2 * rng.standard_normal(K)).astype(np.float32)
    gd, bd = dev(g), dev(b)
    pe_rows, pe_div = 37, 3
    pe = rng.standard_normal((pe_rows, K)).astype(np.float32)
    ped = dev(pe)
    for M in (1, 31, 33, 8191, 8192 + 17, 70001):
        x = (rng.standard_normal((M, K)) + 2.0 * rng.standard_normal((M, 1))).astype(np.float32)
        xd = dev(x)

        def run():
            if split:
                y = torch.full((M, split), float("nan"), device="cuda"); y2 = torch.full((M, N - split), float("nan"), device="cuda")
                capi.check(lib.lfsr_linear_ln_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, ln_cols,
                                                  capi.dev_ptr(ped) if with_pe else None, K, pe_rows, pe_div, capi.dev_ptr(y), split, 0, capi.dev_ptr(y2), N - split, 0, split,
                                                  M, N, capi.stream_ptr()), "linear_ln")
                return np.concatenate([y.cpu().numpy(), y2.cpu().numpy()], axis=1).astype(np.float64)
            y = torch.full((M, N), float("nan"), device="cuda")
            capi.check(lib.lfsr_linear_ln_fwd(capi.dev_ptr(xd), K, 0, K, capi.dev_ptr(wp), capi.dev_ptr(gd), capi.dev_ptr(bd), 1e-5, ln_cols,
                                              capi.dev_ptr(ped) if with_pe else None, K, pe_rows, pe_div, capi.dev_ptr(y), N, 0, None, 0, 0, 0, M, N, capi.stream_ptr()), "linear_ln")
            return y.cpu().numpy().astype(np.float64)
        monkeypatch.delenv("LFSR_LNLIN", raising=False)
        got = run()
        x64 = x.astype(np.float64)
        xp = x64 + (pe[(np.arange(M) // pe_div) % pe_rows] if with_pe else 0.0)
        xn = (xp - xp.mean(-1, keepdims=True)) / np.sqrt(xp.var(-1, keepdims=True) + 1e-5) * g + b
        ref = np.concatenate([xn @ w[:ln_cols].astype(np.float64).T, x64 @ w[ln_cols:].astype(np.float64).T], axis=1)
        assert np.isfinite(got).all() and np.abs(got - ref).max() < 1e-4, (M, np.abs(got - ref).max())
        if ln_cols % 128 == 0 and (split % 128 == 0) and K == 128:
            monkeypatch.setenv("LFSR_LNLIN", "0")
            old = run()
            monkeypatch.delenv("LFSR_LNLIN", raising=False)
            assert np.abs(got - old).max() < 2e-5, (M, np.abs(got - old).max())
