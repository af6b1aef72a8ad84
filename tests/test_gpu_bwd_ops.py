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
