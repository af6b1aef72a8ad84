"""GPU: the operator-level backward entry points of the C ABI (SURVEY 8b export list) through ctypes, against autograd over stock
torch CPU ops of the same layer (what the reference's train.py:256-264 differentiates), and the RCCL all-reduce entry point."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lfsr_amd import capi

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def _vcl(t):       # (n_img, C, h, w) -> (n_img*h*w, C) contiguous on the GPU
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("n_img,h,w", [(50, 32, 32), (7, 13, 40), (3, 5, 6)])
def test_conv3x3_dgrad_wgrad(n_img, h, w):
    g = torch.Generator().manual_seed(n_img)
    x = torch.randn(n_img, 64, h, w, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dy = torch.randn(n_img, 64, h, w, generator=g)
    skip = torch.randn(n_img, 64, h, w, generator=g)
    pre = torch.randn(n_img, 64, h, w, generator=g)             # pre-activation of the layer in front: x = lrelu(pre)
    xin = F.leaky_relu(pre, 0.1).requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    F.conv2d(xin, wr, padding=1).backward(dy)
    dx_ref, dw_ref = xin.grad, wr.grad
    wT = capi.pack_conv_weight_T(wt.cuda())
    dx = capi.conv3x3_dgrad(_vcl(dy), wT, n_img, h, w)
    assert _rel(dx.cpu(), _vcl_cpu(dx_ref)) <= 1e-4
    # through the LeakyReLU in front (mask from the saved activation) plus a skip gradient
    act = F.leaky_relu(pre, 0.1)
    dx2 = capi.conv3x3_dgrad(_vcl(dy), wT, n_img, h, w, res1=_vcl(skip), act=_vcl(act), act_slope=0.1)
    ref2 = dx_ref * torch.where(act > 0, 1.0, 0.1) + skip
    assert _rel(dx2.cpu(), _vcl_cpu(ref2)) <= 1e-4
    for sel in ("", "direct"):          # the Winograd-domain (F(2x2,3x3) adjoint) kernel, default, and the direct-form one
        if sel: os.environ["LFSR_WGRAD3"] = sel
        else: os.environ.pop("LFSR_WGRAD3", None)
        try:
            dw = capi.conv3x3_wgrad(_vcl(dy), _vcl(xin.detach()), n_img, h, w)
            assert _rel(dw.cpu(), dw_ref) <= 1e-4, sel
            dw2 = capi.conv3x3_wgrad(_vcl(dy), _vcl(xin.detach()), n_img, h, w, dw=dw.clone())      # accumulate
            assert _rel(dw2.cpu(), 2 * dw_ref) <= 1e-4, sel
        finally:
            os.environ.pop("LFSR_WGRAD3", None)


def _vcl_cpu(t):
    return t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1])


@pytest.mark.parametrize("M,cin", [(25600, 144), (3000, 144)])
def test_pointwise_dgrad_wgrad(M, cin):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, cin, generator=g)
    wt = (torch.randn(64, cin, generator=g) * 0.1)
    dy = torch.randn(M, 64, generator=g)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    F.linear(xr, wr).backward(dy)
    wT = capi.pack_conv_weight_T(wt.reshape(64, cin, 1, 1).cuda())
    dx = capi.pointwise_dgrad(dy.cuda(), wT, cin)
    assert _rel(dx.cpu(), xr.grad) <= 1e-4
    dxm = capi.pointwise_dgrad(dy.cuda(), wT, cin, act=x.cuda(), act_slope=0.1)
    assert _rel(dxm.cpu(), xr.grad * torch.where(x > 0, 1.0, 0.1)) <= 1e-4
    dw = capi.pointwise_wgrad(dy.cuda(), x.cuda(), 64, cin)
    assert _rel(dw.cpu(), wr.grad) <= 1e-4


def _ps1d(x, f):      # DistgSSR.py:114-131
    B, fC, Hh, Ww = x.shape
    return x.reshape(B, f, fC // f, Hh, Ww).permute(0, 2, 3, 4, 1).reshape(B, fC // f, Hh, Ww * f)


@pytest.mark.parametrize("B,A,h,w", [(2, 5, 32, 32), (1, 5, 6, 9), (2, 3, 8, 8)])
def test_angconv_bwd_vs_autograd(B, A, h, w):
    """lfsr_angconv_bwd against autograd of the reference's AngConv layers (DistgSSR.py:84-90) on stock torch CPU ops: dx (accumulated into a
    given gradient), dW0, dW2; stage-1 activation taken from lfsr_angconv_fwd's tmp output"""
    g = torch.Generator().manual_seed(B * 100 + h)
    x = torch.randn(B, 64, h * A, w * A, generator=g)
    w0 = torch.randn(16, 64, A, A, generator=g) * (1.0 / (64 * A * A) ** 0.5)
    w2 = torch.randn(16 * A * A, 16, 1, 1, generator=g) * 0.25
    dy = torch.randn(B, 16, h * A, w * A, generator=g)
    dx0 = torch.randn(B, 64, h * A, w * A, generator=g)
    xr, w0r, w2r = x.clone().requires_grad_(True), w0.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y_ref = F.pixel_shuffle(F.leaky_relu(F.conv2d(F.leaky_relu(F.conv2d(xr, w0r, stride=A), 0.1), w2r), 0.1), A)
    y_ref.backward(dy)
    xv = capi.nchw_to_vcl(x.cuda(), A, 1)
    out = torch.zeros((xv.shape[0], 16), device="cuda")
    a16 = torch.empty((B * h * w, 16), device="cuda")
    capi.angconv(xv, capi.pack_conv_weight(w0.cuda()), capi.pack_conv_weight(w2.cuda(), perm=1, ch=16), B, A, h, w, 0.1, out, 0, tmp=a16)
    assert float((capi.vcl_to_nchw(out, B, 16, A, h, w, 1).cpu() - y_ref.detach()).abs().max()) <= 1e-4
    dxv = capi.nchw_to_vcl(dx0.cuda(), A, 1)
    dw0, dw2 = capi.angconv_bwd(capi.nchw_to_vcl(dy.cuda(), A, 1), 0, out, 0, xv, a16, w0.cuda(), w2.cuda(), dxv, B, A, h, w)
    assert _rel(capi.vcl_to_nchw(dxv, B, 64, A, h, w, 1).cpu() - dx0, xr.grad) <= 1e-4
    assert _rel(dw0.cpu(), w0r.grad) <= 1e-4
    assert _rel(dw2.cpu(), w2r.grad) <= 1e-4


@pytest.mark.parametrize("B,A,h,w", [(2, 5, 32, 32), (1, 5, 6, 9), (2, 3, 8, 8)])
def test_epiconv_hv_bwd_vs_autograd(B, A, h, w):
    """lfsr_epiconv_hv_bwd against autograd of the reference's EPIConv applied to the tensor and to its transpose with shared weights
    (DistgSSR.py:91-97,108): dx accumulated, dW0 / dW2 summed over both passes; gradients arrive in two channel slices of one VCL buffer"""
    g = torch.Generator().manual_seed(B * 100 + w)
    x = torch.randn(B, 64, h * A, w * A, generator=g)
    w0 = torch.randn(32, 64, 1, A * A, generator=g) * (1.0 / (64 * A * A) ** 0.5)
    w2 = torch.randn(32 * A, 32, 1, 1, generator=g) * 0.18
    dyh = torch.randn(B, 32, h * A, w * A, generator=g)
    dyv = torch.randn(B, 32, h * A, w * A, generator=g)
    dx0 = torch.randn(B, 64, h * A, w * A, generator=g)
    xr, w0r, w2r = x.clone().requires_grad_(True), w0.clone().requires_grad_(True), w2.clone().requires_grad_(True)

    def epi(t):
        e = F.leaky_relu(F.conv2d(t, w0r, stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        return _ps1d(F.leaky_relu(F.conv2d(e, w2r), 0.1), A)
    yh, yv = epi(xr), epi(xr.permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
    (yh * dyh).sum().backward(retain_graph=True)
    (yv * dyv).sum().backward()
    xv = capi.nchw_to_vcl(x.cuda(), A, 1)
    w0p, w2p = capi.pack_conv_weight(w0.cuda()), capi.pack_conv_weight(w2.cuda())
    out = torch.zeros((xv.shape[0], 64), device="cuda")
    eh, ev = torch.empty((B * A * h * w, 32), device="cuda"), torch.empty((B * A * h * w, 32), device="cuda")
    capi.epiconv(xv, w0p, w2p, B, A, h, w, False, 0.1, out, 0, tmp=eh)
    capi.epiconv(xv, w0p, w2p, B, A, h, w, True, 0.1, out, 32, tmp=ev)
    assert float((capi.vcl_to_nchw(out, B, 32, A, h, w, 1, choff=0).cpu() - yh.detach()).abs().max()) <= 1e-4
    assert float((capi.vcl_to_nchw(out, B, 32, A, h, w, 1, choff=32).cpu() - yv.detach()).abs().max()) <= 1e-4
    dyb = torch.zeros((xv.shape[0], 80), device="cuda")             # dLoss/dy_h at channels 8..39, dLoss/dy_v at 48..79 of one buffer
    capi.nchw_to_vcl(dyh.cuda(), A, 1, out=dyb, choff=8)
    capi.nchw_to_vcl(dyv.cuda(), A, 1, out=dyb, choff=48)
    dxv = capi.nchw_to_vcl(dx0.cuda(), A, 1)
    dw0, dw2 = capi.epiconv_hv_bwd(dyb, 8, 48, out, 0, 32, xv, eh, ev, w0.cuda(), w2.cuda(), dxv, B, A, h, w)
    assert _rel(capi.vcl_to_nchw(dxv, B, 64, A, h, w, 1).cpu() - dx0, xr.grad) <= 1e-4
    assert _rel(dw0.cpu(), w0r.grad) <= 1e-4
    assert _rel(dw2.cpu(), w2r.grad) <= 1e-4


def test_upsample_head_dgrad():
    B, A, h, w, s = 2, 5, 8, 8, 4
    g = torch.Generator().manual_seed(3)
    f = torch.randn(B, 64, A * h, A * w, generator=g)                      # SAI-mosaic NCHW features (after MacPI2SAI)
    w0 = torch.randn(64 * s * s, 64, 1, 1, generator=g) * 0.1
    b0 = torch.randn(64 * s * s, generator=g) * 0.1
    w2 = torch.randn(1, 64, 1, 1, generator=g) * 0.1
    dout = torch.randn(B, 1, A * h * s, A * w * s, generator=g)
    fr = f.clone().requires_grad_(True)
    F.conv2d(F.pixel_shuffle(F.conv2d(fr, w0, b0), s), w2).backward(dout)
    lib = capi.load()
    wf = torch.empty(s * s * 64, device="cuda"); bf = torch.empty(s * s, device="cuda")
    w0d, b0d, w2d, doutd = w0.cuda(), b0.cuda(), w2.cuda(), dout.cuda()          # (kept alive across the asynchronous launches)
    capi.check(lib.lfsr_fold_head(capi.dev_ptr(w0d), capi.dev_ptr(b0d), capi.dev_ptr(w2d), capi.dev_ptr(wf), capi.dev_ptr(bf), 64, s,
                                  capi.stream_ptr()), "fold_head")
    npix = B * A * A * h * w
    df = torch.empty(npix, 64, device="cuda"); g16 = torch.empty(npix, 16, device="cuda")
    capi.check(lib.lfsr_upsample_head_dgrad(capi.dev_ptr(doutd), capi.dev_ptr(wf), capi.dev_ptr(df), capi.dev_ptr(g16), B, A, h, w, s,
                                            capi.stream_ptr()), "upsample_head_dgrad")
    got = capi.vcl_to_nchw(df, B, 64, A, h, w, 0).cpu()                   # VCL -> SAI mosaic NCHW
    assert _rel(got, fr.grad) <= 1e-4


def test_rccl_allreduce_single_rank():
    """world size 1 on the one GPU of this box: the RCCL communicator is created through the C ABI and the in-place sum is the identity;
    the N > 1 path is the same call (the driver's multi-GPU leg exercises torch.distributed's RCCL backend by default)."""
    if not capi.load().lfsr_comm_available():
        pytest.skip("librccl not present")
    comm = capi.RcclComm(1, 0)
    t = torch.arange(3581568, dtype=torch.float32, device="cuda") * 1e-3        # the flat DistgSSR gradient bucket's size
    ref = t.clone()
    comm.allreduce_(t)
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    comm.close()
