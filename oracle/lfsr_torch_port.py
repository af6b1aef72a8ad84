"""CPU ORACLE (second form) -- TEST / BASELINE INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

The same restatement as ``lfsr_oracle.py`` but on stock PyTorch CPU ops (``F.conv2d``,
``F.pixel_shuffle``, ``F.interpolate``): i.e. the very ATen/oneDNN kernels the reference's own CPU path
runs (the reference is 100 % Python over stock torch ops).  ``bench.py`` times THIS as
its ``cpu_baseline`` ("port"): it is what a user of the reference gets on the host cores; the numpy
oracle is ~5x slower and would flatter the GPU.  Only tests/ and bench.py's cpu_baseline leg import it.
Pinned against the reference's golden vectors in tests/test_oracle_vs_golden.py.
"""
import torch
import torch.nn.functional as F


def sai2macpi(x, A):
    """DistgSSR.py:145-155."""
    B, C, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    return x.reshape(B, C, A, h, A, w).permute(0, 1, 3, 2, 5, 4).reshape(B, C, h * A, w * A)


def macpi2sai(x, A):
    """DistgSSR.py:134-142."""
    B, C, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    return x.reshape(B, C, h, A, w, A).permute(0, 1, 3, 2, 5, 4).reshape(B, C, A * h, A * w)


def pixel_shuffle1d(x, f):
    """DistgSSR.py:114-131."""
    B, fC, H, W = x.shape
    C = fC // f
    return x.reshape(B, f, C, H, W).permute(0, 2, 3, 4, 1).reshape(B, C, H, W * f)


def distg_block(x, sd, pre, A):
    """DisentgBlock.forward DistgSSR.py:104-111."""
    lr = lambda t: F.leaky_relu(t, 0.1)
    spa = lr(F.conv2d(x, sd[pre + "SpaConv.0.weight"], dilation=A, padding=A))
    spa = lr(F.conv2d(spa, sd[pre + "SpaConv.2.weight"], dilation=A, padding=A))
    ang = lr(F.conv2d(x, sd[pre + "AngConv.0.weight"], stride=A))
    ang = F.pixel_shuffle(lr(F.conv2d(ang, sd[pre + "AngConv.2.weight"])), A)

    def epi(t):
        e = lr(F.conv2d(t, sd[pre + "EPIConv.0.weight"], stride=(1, A), padding=(0, A * (A - 1) // 2)))
        return pixel_shuffle1d(lr(F.conv2d(e, sd[pre + "EPIConv.2.weight"])), A)
    epih = epi(x)
    epiv = epi(x.permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
    buf = torch.cat((spa, ang, epih, epiv), dim=1)
    buf = lr(F.conv2d(buf, sd[pre + "fuse.0.weight"]))
    return F.conv2d(buf, sd[pre + "fuse.2.weight"], dilation=A, padding=A) + x


@torch.no_grad()
def distgssr_forward(x, sd, A, s, n_group=4, n_block=4):
    """get_model.forward DistgSSR.py:29-36.  x, sd: torch CPU tensors."""
    x_up = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
    buf0 = F.conv2d(sai2macpi(x, A), sd["init_conv.weight"], dilation=A, padding=A)
    buf = buf0
    for g in range(n_group):
        gin = buf
        for b in range(n_block):
            buf = distg_block(buf, sd, f"disentg.Group.{g}.Block.{b}.", A)
        buf = F.conv2d(buf, sd[f"disentg.Group.{g}.conv.weight"], dilation=A, padding=A) + gin
    buf = F.conv2d(buf, sd["disentg.conv.weight"], dilation=A, padding=A) + buf0
    up = F.conv2d(macpi2sai(buf, A), sd["upsample.0.weight"], sd["upsample.0.bias"])
    up = F.conv2d(F.pixel_shuffle(up, s), sd["upsample.2.weight"])
    return up + x_up
