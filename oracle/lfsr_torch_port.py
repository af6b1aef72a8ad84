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


def distg_block(x, sd, pre, A, rec=None, force=None):
    """DisentgBlock.forward DistgSSR.py:104-111.  rec: optional dict that receives the sign (> 0) of every LeakyReLU output of the block
    (= the LeakyReLU' masks autograd applies), keyed pre + {S1, S2, A1, A2, EH1, EH2, EV1, EV2, FZ} in the reference's own layouts.
    force: optional dict of such masks to APPLY instead of the sign test (y = t where mask else 0.1 t): the same graph with another
    implementation's discrete LeakyReLU decisions -- what separates mask flips at pre-activations within round-off of zero from arithmetic error."""
    def lr(t, tag=None):
        if force is not None and tag:
            y = torch.where(force[pre + tag], t, 0.1 * t)
        else:
            y = F.leaky_relu(t, 0.1)
        if rec is not None and tag:
            rec[pre + tag] = (y.detach() > 0)
        return y
    spa = lr(F.conv2d(x, sd[pre + "SpaConv.0.weight"], dilation=A, padding=A), "S1")
    spa = lr(F.conv2d(spa, sd[pre + "SpaConv.2.weight"], dilation=A, padding=A), "S2")
    ang = lr(F.conv2d(x, sd[pre + "AngConv.0.weight"], stride=A), "A1")
    ang = F.pixel_shuffle(lr(F.conv2d(ang, sd[pre + "AngConv.2.weight"]), "A2"), A)

    def epi(t, tag):
        e = lr(F.conv2d(t, sd[pre + "EPIConv.0.weight"], stride=(1, A), padding=(0, A * (A - 1) // 2)), tag + "1")
        return pixel_shuffle1d(lr(F.conv2d(e, sd[pre + "EPIConv.2.weight"]), tag + "2"), A)
    epih = epi(x, "EH")
    epiv = epi(x.permute(0, 1, 3, 2).contiguous(), "EV").permute(0, 1, 3, 2)
    buf = torch.cat((spa, ang, epih, epiv), dim=1)
    buf = lr(F.conv2d(buf, sd[pre + "fuse.0.weight"]), "FZ")
    return F.conv2d(buf, sd[pre + "fuse.2.weight"], dilation=A, padding=A) + x


def distgssr_forward_graph(x, sd, A, s, n_group=4, n_block=4, rec=None, force=None):
    """get_model.forward DistgSSR.py:29-36 with autograd left on (the gradient parity tests differentiate THIS with
    ``sd`` holding leaf tensors that require grad: the fp32 CPU forward + backward the reference's train.py:256-264 runs)."""
    x_up = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
    buf0 = F.conv2d(sai2macpi(x, A), sd["init_conv.weight"], dilation=A, padding=A)
    buf = buf0
    for g in range(n_group):
        gin = buf
        for b in range(n_block):
            buf = distg_block(buf, sd, f"disentg.Group.{g}.Block.{b}.", A, rec, force)
        buf = F.conv2d(buf, sd[f"disentg.Group.{g}.conv.weight"], dilation=A, padding=A) + gin
    buf = F.conv2d(buf, sd["disentg.conv.weight"], dilation=A, padding=A) + buf0
    up = F.conv2d(macpi2sai(buf, A), sd["upsample.0.weight"], sd["upsample.0.bias"])
    up = F.conv2d(F.pixel_shuffle(up, s), sd["upsample.2.weight"])
    return up + x_up


@torch.no_grad()
def distgssr_forward(x, sd, A, s, n_group=4, n_block=4):
    """Inference form (``torch.no_grad()``, train.py:300-313).  x, sd: torch CPU tensors."""
    return distgssr_forward_graph(x, sd, A, s, n_group, n_block)


# ----------------------------------------------------------------------------------------------
# EPIT / LFT / LF_InterNet on stock torch CPU ops (timed as cpu_baseline lines for configs 3, 5, 1)
# ----------------------------------------------------------------------------------------------

def _conv133(x, w):
    """nn.Conv3d(k=(1,3,3), pad=(0,1,1), bias=False): EPIT.py:24-32,136-142 / LFT.py:36-46."""
    return F.conv3d(x, w, padding=(0, 1, 1))


def _mha(q, k, v, in_w, out_w, nheads, mask=None):
    """nn.MultiheadAttention(need_weights=False) forward, (L,N,E), no biases: in-proj slices, SDPA, out-proj (EPIT.py:118-122,
    LFT.py:195-199,238-241)."""
    L, N, E = q.shape
    hd = E // nheads
    Q = F.linear(q, in_w[:E]).reshape(L, N * nheads, hd).transpose(0, 1)
    K = F.linear(k, in_w[E:2 * E]).reshape(L, N * nheads, hd).transpose(0, 1)
    V = F.linear(v, in_w[2 * E:]).reshape(L, N * nheads, hd).transpose(0, 1)
    O = F.scaled_dot_product_attention(Q, K, V, attn_mask=mask)
    return F.linear(O.transpose(0, 1).reshape(L, N, E), out_w)


def _window_mask(h, w, kh_l, kh_r, kw_l, kw_r, wclamp):
    """additive (h*w, h*w) float mask, 0 inside the window / -inf outside (EPIT.py:93-108; LFT.py:161-174 with wclamp = h)."""
    i = torch.arange(h).view(h, 1, 1, 1); j = torch.arange(w).view(1, w, 1, 1)
    ii = torch.arange(h).view(1, 1, h, 1); jj = torch.arange(w).view(1, 1, 1, w)
    ok = (ii >= i - kh_l) & (ii < i + kh_r) & (jj >= j - kw_l) & (jj < torch.clamp(j + kw_r, max=wclamp))
    m = torch.full((h, w, h, w), float("-inf"))
    m[ok] = 0.0
    return m.reshape(h * w, h * w)


def _ffn(tok, sd, pre):
    ff = F.layer_norm(tok, tok.shape[-1:], sd[pre + "feed_forward.0.weight"], sd[pre + "feed_forward.0.bias"])
    return F.linear(F.relu(F.linear(ff, sd[pre + "feed_forward.1.weight"])), sd[pre + "feed_forward.4.weight"]) + tok


def _epit_basic_trans(buf, sd, pre, mf):
    """BasicTrans.forward EPIT.py:110-128.  buf (b,c,n,v,w)."""
    b, c, n, v, w = buf.shape
    mask = _window_mask(v, w, mf[0] // 2, mf[0] - mf[0] // 2, mf[1] // 2, mf[1] - mf[1] // 2, w)
    tok = buf.permute(3, 4, 0, 2, 1).reshape(v * w, b * n, c)
    tok = F.linear(tok, sd[pre + "linear_in.weight"])
    tn = F.layer_norm(tok, tok.shape[-1:], sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    tok = _mha(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], 8, mask) + tok
    tok = F.linear(_ffn(tok, sd, pre), sd[pre + "linear_out.weight"])
    return tok.reshape(v, w, b, n, -1).permute(2, 4, 3, 0, 1)


def _up_tail(buf, sd, A, h, w, s, skip):
    B, c = buf.shape[:2]
    mosaic = buf.reshape(B, c, A, A, h, w).permute(0, 1, 2, 4, 3, 5).reshape(B, c, A * h, A * w)
    up = F.pixel_shuffle(F.conv2d(mosaic, sd["upsampling.0.weight"]), s)
    return F.conv2d(F.leaky_relu(up, 0.2), sd["upsampling.3.weight"], padding=1) + skip


def _bicubic_views(x, A, h, w, s):
    B = x.shape[0]
    lr = x.reshape(B, 1, A, h, A, w).permute(0, 1, 2, 4, 3, 5)
    sr = F.interpolate(lr.reshape(B * A * A, 1, h, w), scale_factor=s, mode="bicubic", align_corners=False)
    sr = sr.reshape(B, 1, A, A, h * s, w * s).permute(0, 1, 2, 4, 3, 5).reshape(B, 1, A * h * s, A * w * s)
    return lr.reshape(B, 1, A * A, h, w), sr


def _init_feats(v, sd):
    buf = _conv133(v, sd["conv_init0.0.weight"])
    t = buf
    for i in (0, 2, 4):
        t = F.leaky_relu(_conv133(t, sd[f"conv_init.{i}.weight"]), 0.2)
    return t + buf


@torch.no_grad()
def epit_forward(x, sd, A, s):
    """get_model.forward EPIT.py:51-71 (AltFilter :144-161: one epi_trans and one conv shared by both passes, the block INPUT
    added after each pass)."""
    B, _, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    v, sr = _bicubic_views(x, A, h, w, s)
    buf = _init_feats(v, sd)
    t = buf
    nblk = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("altblock."))
    mf = [A * 2, 11]
    for i in range(nblk):
        pre = f"altblock.{i}."
        b, c = t.shape[:2]
        shortcut = t

        def conv(z):
            z = F.leaky_relu(_conv133(z, sd[pre + "conv.0.weight"]), 0.2)
            z = F.leaky_relu(_conv133(z, sd[pre + "conv.2.weight"]), 0.2)
            return _conv133(z, sd[pre + "conv.4.weight"])
        z = t.reshape(b, c, A, A, h, w).permute(0, 1, 3, 5, 2, 4).reshape(b, c, A * w, A, h)
        z = _epit_basic_trans(z, sd, pre + "epi_trans.", mf)
        z = z.reshape(b, c, A, w, A, h).permute(0, 1, 4, 2, 5, 3).reshape(b, c, A * A, h, w)
        t = conv(z) + shortcut
        z = t.reshape(b, c, A, A, h, w).permute(0, 1, 2, 4, 3, 5).reshape(b, c, A * h, A, w)
        z = _epit_basic_trans(z, sd, pre + "epi_trans.", mf)
        z = z.reshape(b, c, A, h, A, w).permute(0, 1, 2, 4, 3, 5).reshape(b, c, A * A, h, w)
        t = conv(z) + shortcut
    return _up_tail(t + buf, sd, A, h, w, s, sr)


def _lft_pe(lengths, dim, temperature=10000):
    """PositionEncoding.forward LFT.py:106-130 (sin of the even columns then cos of the odd ones, concatenated)."""
    grid = torch.arange(dim, dtype=torch.float64)
    grid = temperature ** (2 * torch.div(grid, 2, rounding_mode="floor") / dim)
    out = []
    for n in lengths:
        pos = torch.arange(n, dtype=torch.float64).view(-1, 1) / grid
        out.append(torch.cat([pos[:, 0::2].sin(), pos[:, 1::2].cos()], dim=1).float())
    return out


@torch.no_grad()
def lft_forward(x, sd, A, s):
    """get_model.forward LFT.py:67-98 (AngTrans :233-246, SpaTrans :188-203 incl. the h-for-w clamp of gen_mask :161-174)."""
    B, _, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    v, up = _bicubic_views(x, A, h, w, s)
    buf = _init_feats(v, sd)
    c = buf.shape[1]
    ph, pw, pa = _lft_pe([h, w, A * A], c)
    spa_pos = ((ph[:, None, :] + pw[None, :, :]) / 2).permute(2, 0, 1).reshape(1, c, 1, h, w)
    mask = _window_mask(h, w, 2, 3, 2, 3, h)
    t = buf
    nblk = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("altblock."))
    for i in range(nblk):
        pre = f"altblock.{i}.ang_trans."
        b, _, a, _, _ = t.shape
        tok = t.permute(2, 0, 3, 4, 1).reshape(a, b * h * w, c)
        tn = F.layer_norm(tok + pa.reshape(a, 1, c), (c,), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
        tok = _mha(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], 8) + tok
        tok = _ffn(tok, sd, pre)
        t = tok.reshape(a, b, h, w, c).permute(1, 4, 0, 2, 3)
        pre = f"altblock.{i}.spa_trans."
        wm = sd[pre + "MLP.weight"]

        def sai2token(z):   # F.unfold(k3, pad 1) + Linear(576 -> 128), LFT.py:176-182
            n = z.shape[0] * z.shape[2]
            u = F.unfold(z.permute(0, 2, 1, 3, 4).reshape(n, c, h, w), kernel_size=3, padding=1)   # (n, 576, h w)
            return F.linear(u.permute(2, 0, 1), wm)
        tok = sai2token(t)
        tn = F.layer_norm(tok + sai2token(spa_pos), tok.shape[-1:], sd[pre + "norm.weight"], sd[pre + "norm.bias"])
        tok = _mha(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], 8, mask) + tok
        tok = _ffn(tok, sd, pre)
        z = tok.reshape(h, w, b, a, -1).permute(2, 4, 3, 0, 1)
        t = F.conv3d(z, sd[pre + "linear.0.weight"])
    return _up_tail(t + buf, sd, A, h, w, s, up)


@torch.no_grad()
def internet_forward(x, sd, A, s, n_groups=4, n_layers=4):
    """get_model.forward LF_InterNet.py:33-41 (make_chains :44-67, BottleNeck :107-124, ReconBlock :127-141)."""
    m = sai2macpi(x, A)
    xa = F.conv2d(m, sd["AngFE.0.weight"], stride=A)
    xs = F.conv2d(m, sd["SpaFE.0.weight"], dilation=A, padding=A)
    ba, bs = xa, xs
    oa_l, os_l = [], []
    for g in range(n_groups):
        for l in range(n_layers):
            p = f"CascadeInterBlock.body.{g}.chained_layers.{l}."
            ang2 = F.relu(F.conv2d(bs, sd[p + "Spa2Ang.weight"], stride=A))
            spa2 = F.pixel_shuffle(F.conv2d(ba, sd[p + "Ang2Spa.0.weight"]), A)
            oa = F.relu(F.conv2d(torch.cat((ba, ang2), 1), sd[p + "AngConvSq.weight"])) + ba
            os_ = F.relu(F.conv2d(torch.cat((bs, spa2), 1), sd[p + "SpaConvSq.weight"], dilation=A, padding=A)) + bs
            ba, bs = oa, os_
        oa_l.append(ba)
        os_l.append(bs)
    a = F.relu(F.conv2d(torch.cat(oa_l, 1), sd["BottleNeck.AngBottle.weight"]))
    cs = torch.cat((torch.cat(os_l, 1), F.pixel_shuffle(F.conv2d(a, sd["BottleNeck.Ang2Spa.0.weight"]), A)), 1)
    out = F.relu(F.conv2d(cs, sd["BottleNeck.SpaBottle.weight"], dilation=A, padding=A)) + xs
    pre = F.conv2d(out, sd["ReconBlock.PreConv.weight"], dilation=A, padding=A)
    return F.conv2d(F.pixel_shuffle(macpi2sai(pre, A), s), sd["ReconBlock.FinalConv.weight"])
