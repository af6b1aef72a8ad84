"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

A plain-numpy restatement of the reference's light-field SR hot path (BasicLFSR fork at
``/root/reference``), written in the *reference's own* formulation (NCHW tensors, macro-pixel layout,
dilated convolutions, explicit rearranges) so that it is an independent check of the HIP path, which
computes the same functions in a different (view-major, channel-last) formulation.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module -- as the checker / the timed CPU baseline, never as a fallback for the product.

Pinned: every function here is checked against golden vectors produced by importing and running the
reference itself in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py``).  The reference holds no tests or fixtures of its own for this path
(SURVEY.md section 4).

All arithmetic runs in ``dtype`` (float64 by default: a sharper checker than the reference's own
fp32; pass ``np.float32`` to mirror the reference's precision, which is what the CPU baseline times).
Each function cites the reference file:line it restates.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# a1-a7: integer-indexing primitives (bit-exact)
# ----------------------------------------------------------------------------------------------


def sai2macpi(x, A):
    """``SAI2MacPI`` DistgSSR.py:145-155 (dup LF_InterNet.py:155-165).
    out[b,c,y*A+u,x*A+v] = in[b,c,u*h+y,v*w+x]."""
    B, C, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    return np.ascontiguousarray(x.reshape(B, C, A, h, A, w).transpose(0, 1, 3, 2, 5, 4).reshape(B, C, h * A, w * A))


def macpi2sai(x, A):
    """``MacPI2SAI`` DistgSSR.py:134-142.  out[b,c,u*h+y,v*w+x] = in[b,c,y*A+u,x*A+v]."""
    B, C, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    return np.ascontiguousarray(x.reshape(B, C, h, A, w, A).transpose(0, 1, 3, 2, 5, 4).reshape(B, C, A * h, A * w))


def pixel_shuffle(x, r):
    """``nn.PixelShuffle(r)`` as used at DistgSSR.py:26,89; EPIT.py:46; LFT.py:54; LF_InterNet.py:51,114,132.
    out[b,c,y*r+i,x*r+j] = in[b,c*r*r+i*r+j,y,x]."""
    B, Crr, H, W = x.shape
    C = Crr // (r * r)
    return np.ascontiguousarray(x.reshape(B, C, r, r, H, W).transpose(0, 1, 4, 2, 5, 3).reshape(B, C, H * r, W * r))


def pixel_shuffle1d(x, f):
    """``PixelShuffle1D`` DistgSSR.py:114-131 (factor-major channels).  out[b,c,y,x*f+k] = in[b,k*C+c,y,x]."""
    B, fC, H, W = x.shape
    C = fC // f
    return np.ascontiguousarray(x.reshape(B, f, C, H, W).transpose(0, 2, 3, 4, 1).reshape(B, C, H, W * f))


def _sym(i, n):
    """Index into a length-n axis under edge-including mirror extension of period 2n."""
    i = np.mod(i, 2 * n)
    return np.where(i < n, i, 2 * n - 1 - i)


def image_extend(im, bdr):
    """``ImageExtend`` utils/utils.py:137-149: the 3x3 flip mosaic cropped to
    [h-bdr0 : 2h+bdr1, w-bdr2 : 2w+bdr3]."""
    h, w = im.shape[-2:]
    ys = _sym(np.arange(-bdr[0], h + bdr[1]), h)
    xs = _sym(np.arange(-bdr[2], w + bdr[3]), w)
    return np.ascontiguousarray(im[..., ys[:, None], xs[None, :]])


def lf_divide(data, A, P, S):
    """``LFdivide`` utils/utils.py:152-166.  data (A*h0, A*w0) -> (numU, numV, A*P, A*P)."""
    h0, w0 = data.shape[0] // A, data.shape[1] // A
    d = data.reshape(A, h0, A, w0).transpose(0, 2, 1, 3).reshape(A * A, 1, h0, w0)
    bdr = (P - S) // 2
    numU = (h0 + bdr * 2 - 1) // S
    numV = (w0 + bdr * 2 - 1) // S
    pad = image_extend(d, [bdr, bdr + S - 1, bdr, bdr + S - 1])[:, 0]          # (A*A, hp, wp)
    out = np.empty((numU, numV, A, P, A, P), dtype=data.dtype)
    for n1 in range(numU):                                                      # F.unfold(k=P, stride=S)
        for n2 in range(numV):
            blk = pad[:, n1 * S:n1 * S + P, n2 * S:n2 * S + P].reshape(A, A, P, P)
            out[n1, n2] = blk.transpose(0, 2, 1, 3)
    return out.reshape(numU, numV, A * P, A * P)


def lf_integrate(sub, A, pz, stride, h, w):
    """``LFintegrate`` utils/utils.py:169-178: centre-crop stride x stride of every patch, tile, crop."""
    if sub.ndim == 4:
        n1, n2 = sub.shape[:2]
        sub = sub.reshape(n1, n2, A, pz, A, pz).transpose(0, 1, 2, 4, 3, 5)
    n1, n2 = sub.shape[:2]
    bdr = (pz - stride) // 2
    o = sub[:, :, :, :, bdr:bdr + stride, bdr:bdr + stride]
    o = o.transpose(2, 3, 0, 4, 1, 5).reshape(A, A, n1 * stride, n2 * stride)
    return np.ascontiguousarray(o[:, :, :h, :w])


def lf_crop_tiles(sub, A, pz, stride):
    """The per-patch half of ``LFintegrate`` (utils/utils.py:169-175): sub (n, A*pz, A*pz) [or (n,1,A*pz,A*pz)] -> (n, A, A, stride, stride),
    the centre stride x stride of every view -- everything ``LFintegrate`` keeps of a patch."""
    n = sub.shape[0]
    v = sub.reshape(n, A, pz, A, pz)
    bdr = (pz - stride) // 2
    return np.ascontiguousarray(v[:, :, bdr:bdr + stride, :, bdr:bdr + stride].transpose(0, 1, 3, 2, 4))


def lf_place_tiles(tiles, out, A, numU, numV, first, stride):
    """The assembling half (utils/utils.py:176-178): tiles of patches [first, first+count) of the row-major (numU, numV) list go to
    out[a1, a2, n1*stride + y, n2*stride + x], cropped at out's (h, w).  In place."""
    h, w = out.shape[2:]
    for k in range(tiles.shape[0]):
        n1, n2 = divmod(first + k, numV)
        y0, x0 = n1 * stride, n2 * stride
        hh, ww = min(stride, h - y0), min(stride, w - x0)
        if hh > 0 and ww > 0:
            out[:, :, y0:y0 + hh, x0:x0 + ww] = tiles[k, :, :, :hh, :ww]
    return out


# ----------------------------------------------------------------------------------------------
# floating-point building blocks (stock PyTorch ops the reference calls)
# ----------------------------------------------------------------------------------------------


def conv2d(x, w, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """``nn.Conv2d`` forward (cross-correlation), NCHW.  Tap-by-tap GEMM accumulation."""
    B, C, H, W = x.shape
    O, Cw, kh, kw = w.shape
    assert C == Cw
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=x.dtype)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    out = np.zeros((B, O, Ho * Wo), dtype=x.dtype)
    wt = np.ascontiguousarray(w.transpose(2, 3, 0, 1))                 # (kh, kw, O, C): contiguous GEMM operands
    for i in range(kh):
        for j in range(kw):
            sl = xp[:, :, i * dh:i * dh + sh * (Ho - 1) + 1:sh, j * dw:j * dw + sw * (Wo - 1) + 1:sw]
            out += np.matmul(wt[i, j], np.ascontiguousarray(sl).reshape(B, C, Ho * Wo))
    out = out.reshape(B, O, Ho, Wo)
    if bias is not None:
        out = out + bias.reshape(1, O, 1, 1)
    return out


def leaky_relu(x, slope):
    return np.where(x >= 0, x, x * np.asarray(slope, dtype=x.dtype))


def _lin_coords(n_in, scale, dtype):
    """PyTorch area_pixel_compute_source_index, align_corners=False, scale_factor given."""
    dst = np.arange(n_in * scale, dtype=np.float64)
    return (dst + 0.5) / scale - 0.5


def interp_bilinear(x, s):
    """``F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)`` (DistgSSR.py:30)."""
    def axis(n):
        src = np.maximum(_lin_coords(n, s, x.dtype), 0.0)
        i0 = np.floor(src).astype(np.int64)
        i0 = np.minimum(i0, n - 1)
        i1 = np.minimum(i0 + 1, n - 1)
        l1 = (src - i0).astype(x.dtype)
        return i0, i1, (1 - l1).astype(x.dtype), l1
    H, W = x.shape[-2:]
    y0, y1, wy0, wy1 = axis(H)
    x0, x1, wx0, wx1 = axis(W)
    rows = x[..., y0, :] * wy0[:, None] + x[..., y1, :] * wy1[:, None]
    return rows[..., :, x0] * wx0 + rows[..., :, x1] * wx1


def interp_bicubic(x, s):
    """``F.interpolate(..., mode='bicubic', align_corners=False)`` (EPIT.py:167, LFT.py:269): a=-0.75,
    border indices clamped."""
    a = -0.75

    def cc1(t):
        return ((a + 2) * t - (a + 3)) * t * t + 1

    def cc2(t):
        return ((a * t - 5 * a) * t + 8 * a) * t - 4 * a

    def axis(n):
        src = _lin_coords(n, s, x.dtype)
        i0 = np.floor(src)
        t = src - i0
        i0 = i0.astype(np.int64)
        idx = np.stack([np.clip(i0 + k, 0, n - 1) for k in (-1, 0, 1, 2)])
        wts = np.stack([cc2(t + 1), cc1(t), cc1(1 - t), cc2(2 - t)]).astype(x.dtype)
        return idx, wts
    H, W = x.shape[-2:]
    iy, wy = axis(H)
    ix, wx = axis(W)
    rows = sum(x[..., iy[k], :] * wy[k][:, None] for k in range(4))
    return sum(rows[..., :, ix[k]] * wx[k] for k in range(4))


def linear(x, w, b=None):
    y = np.matmul(x, w.T)
    return y if b is None else y + b


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def mha(q, k, v, in_proj_w, out_proj_w, nheads, mask=None):
    """``nn.MultiheadAttention`` forward, (L,N,E) layout, no biases, additive float mask (L,L)."""
    L, N, E = q.shape
    hd = E // nheads
    wq, wk, wv = in_proj_w[:E], in_proj_w[E:2 * E], in_proj_w[2 * E:]
    Q = linear(q, wq).reshape(L, N * nheads, hd).transpose(1, 0, 2)
    K = linear(k, wk).reshape(L, N * nheads, hd).transpose(1, 0, 2)
    V = linear(v, wv).reshape(L, N * nheads, hd).transpose(1, 0, 2)
    S = np.matmul(Q, K.transpose(0, 2, 1)) / np.sqrt(np.asarray(hd, dtype=q.dtype))
    if mask is not None:
        S = S + mask
    S = S - S.max(-1, keepdims=True)
    P = np.exp(S)
    P = P / P.sum(-1, keepdims=True)
    O = np.matmul(P, V).transpose(1, 0, 2).reshape(L, N, E)
    return linear(O, out_proj_w)


# ----------------------------------------------------------------------------------------------
# DistgSSR (model/SR/DistgSSR.py)
# ----------------------------------------------------------------------------------------------


def _cast(sd, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}


def distg_block(x, sd, pre, A, taps=None):
    """``DisentgBlock.forward`` DistgSSR.py:104-111 on the MacPI tensor x (B,64,A*h,A*w)."""
    d = (A, A)
    spa = leaky_relu(conv2d(x, sd[pre + "SpaConv.0.weight"], dilation=d, padding=d), 0.1)
    spa = leaky_relu(conv2d(spa, sd[pre + "SpaConv.2.weight"], dilation=d, padding=d), 0.1)
    ang = leaky_relu(conv2d(x, sd[pre + "AngConv.0.weight"], stride=d), 0.1)
    ang = leaky_relu(conv2d(ang, sd[pre + "AngConv.2.weight"]), 0.1)
    ang = pixel_shuffle(ang, A)

    def epi(t):
        e = leaky_relu(conv2d(t, sd[pre + "EPIConv.0.weight"], stride=(1, A), padding=(0, A * (A - 1) // 2)), 0.1)
        e = leaky_relu(conv2d(e, sd[pre + "EPIConv.2.weight"]), 0.1)
        return pixel_shuffle1d(e, A)
    epih = epi(x)
    epiv_t = epi(np.ascontiguousarray(x.transpose(0, 1, 3, 2)))
    epiv = epiv_t.transpose(0, 1, 3, 2)
    buf = np.concatenate((spa, ang, epih, epiv), axis=1)
    buf = leaky_relu(conv2d(buf, sd[pre + "fuse.0.weight"]), 0.1)
    buf = conv2d(buf, sd[pre + "fuse.2.weight"], dilation=d, padding=d)
    if taps is not None:
        taps.update(spa=spa, ang=ang, epih=epih, epiv_t=epiv_t)
    return buf + x


def distgssr_forward(x, sd, A, s, dtype=np.float64, taps=None):
    """``get_model.forward`` DistgSSR.py:29-36.  x (B,1,A*h,A*w) SAI mosaic -> (B,1,A*h*s,A*w*s)."""
    sd = _cast(sd, dtype)
    x = np.asarray(x, dtype=dtype)
    d = (A, A)
    x_up = interp_bilinear(x, s)
    m = sai2macpi(x, A)
    buf0 = conv2d(m, sd["init_conv.weight"], dilation=d, padding=d)
    if taps is not None:
        taps["init_conv"] = buf0
    buf = buf0
    n_group = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("disentg.Group."))
    for g in range(n_group):
        gin = buf
        n_block = 1 + max(int(k.split(".")[4]) for k in sd if k.startswith(f"disentg.Group.{g}.Block."))
        for b in range(n_block):
            t = {} if (taps is not None and g == 0 and b == 0) else None
            buf = distg_block(buf, sd, f"disentg.Group.{g}.Block.{b}.", A, t)
            if t is not None:
                taps.update({"b0_" + k: v for k, v in t.items()})
                taps["b0_out"] = buf
        buf = conv2d(buf, sd[f"disentg.Group.{g}.conv.weight"], dilation=d, padding=d) + gin
        if taps is not None and g == 0:
            taps["g0_out"] = buf
    buf = conv2d(buf, sd["disentg.conv.weight"], dilation=d, padding=d) + buf0
    if taps is not None:
        taps["disentg_out"] = buf
    sai = macpi2sai(buf, A)
    up = conv2d(sai, sd["upsample.0.weight"], sd["upsample.0.bias"])
    up = pixel_shuffle(up, s)
    up = conv2d(up, sd["upsample.2.weight"])
    return up + x_up


def l1_loss(sr, hr):
    """``get_loss`` DistgSSR.py:158-166 (``nn.L1Loss`` mean)."""
    return np.abs(sr - hr).mean()


def psnr(a, b):
    """10 log10(1 / MSE), data_range 1.0 (formula of utils/utils.py:109 as skimage computes it)."""
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


# ----------------------------------------------------------------------------------------------
# EPIT (model/SR/EPIT.py)
# ----------------------------------------------------------------------------------------------


def conv3d_133(x, w):
    """``nn.Conv3d(kernel=(1,3,3), padding=(0,1,1), bias=False)`` EPIT.py:24-32,136-142 / LFT.py:36-46:
    a per-view 3x3 conv.  x (B,C,N,h,w), w (O,C,1,3,3)."""
    B, C, N, h, wd = x.shape
    y = conv2d(x.transpose(0, 2, 1, 3, 4).reshape(B * N, C, h, wd), w[:, :, 0], padding=(1, 1))
    return y.reshape(B, N, -1, h, wd).transpose(0, 2, 1, 3, 4)


def epit_gen_mask(h, w, k_h, k_w, dtype):
    """``BasicTrans.gen_mask`` EPIT.py:93-108: additive (h*w, h*w) mask, 0 inside the window, -inf outside."""
    khl, kwl = k_h // 2, k_w // 2
    khr, kwr = k_h - khl, k_w - kwl
    m = np.full((h, w, h, w), -np.inf, dtype=dtype)
    for i in range(h):
        for j in range(w):
            m[i, j, max(0, i - khl):min(h, i + khr), max(0, j - kwl):min(w, j + kwr)] = 0.0
    return m.reshape(h * w, h * w)


def epit_basic_trans(buf, sd, pre, mask_field, nheads=8):
    """``BasicTrans.forward`` EPIT.py:110-128.  buf (b,c,n,v,w) -> same shape."""
    b, c, n, v, w = buf.shape
    mask = epit_gen_mask(v, w, mask_field[0], mask_field[1], buf.dtype)
    tok = buf.transpose(3, 4, 0, 2, 1).reshape(v * w, b * n, c)                   # (v w) (b n) c
    tok = linear(tok, sd[pre + "linear_in.weight"])
    tn = layer_norm(tok, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    tok = mha(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], nheads, mask) + tok
    ff = layer_norm(tok, sd[pre + "feed_forward.0.weight"], sd[pre + "feed_forward.0.bias"])
    ff = np.maximum(linear(ff, sd[pre + "feed_forward.1.weight"]), 0)
    tok = linear(ff, sd[pre + "feed_forward.4.weight"]) + tok
    tok = linear(tok, sd[pre + "linear_out.weight"])
    return tok.reshape(v, w, b, n, -1).transpose(2, 4, 3, 0, 1)                   # b c n v w


def epit_alt_filter(buf, sd, pre, A):
    """``AltFilter.forward`` EPIT.py:144-161.  buf (b,c,A*A,h,w).  One epi_trans and one conv shared by both passes,
    and the ORIGINAL input added after each pass."""
    b, c, _, h, w = buf.shape
    shortcut = buf
    mf = [A * 2, 11]

    def conv(t):
        t = leaky_relu(conv3d_133(t, sd[pre + "conv.0.weight"]), 0.2)
        t = leaky_relu(conv3d_133(t, sd[pre + "conv.2.weight"]), 0.2)
        return conv3d_133(t, sd[pre + "conv.4.weight"])
    # horizontal: 'b c (u v) h w -> b c (v w) u h'
    t = buf.reshape(b, c, A, A, h, w).transpose(0, 1, 3, 5, 2, 4).reshape(b, c, A * w, A, h)
    t = epit_basic_trans(t, sd, pre + "epi_trans.", mf)
    t = t.reshape(b, c, A, w, A, h).transpose(0, 1, 4, 2, 5, 3).reshape(b, c, A * A, h, w)
    buf = conv(t) + shortcut
    # vertical: 'b c (u v) h w -> b c (u h) v w'
    t = buf.reshape(b, c, A, A, h, w).transpose(0, 1, 2, 4, 3, 5).reshape(b, c, A * h, A, w)
    t = epit_basic_trans(t, sd, pre + "epi_trans.", mf)
    t = t.reshape(b, c, A, h, A, w).transpose(0, 1, 2, 4, 3, 5).reshape(b, c, A * A, h, w)
    return conv(t) + shortcut


def epit_forward(x, sd, A, s, dtype=np.float64, taps=None):
    """``get_model.forward`` EPIT.py:51-71.  x (B,1,A*h,A*w) -> (B,1,A*h*s,A*w*s)."""
    sd = _cast(sd, dtype)
    x = np.asarray(x, dtype=dtype)
    B, _, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    lr = x.reshape(B, 1, A, h, A, w).transpose(0, 1, 2, 4, 3, 5)                  # b c u v h w
    sr = interp_bicubic(lr.reshape(B * A * A, 1, h, w), s).reshape(B, 1, A, A, h * s, w * s)
    sr = sr.transpose(0, 1, 2, 4, 3, 5).reshape(B, 1, A * h * s, A * w * s)
    v = lr.reshape(B, 1, A * A, h, w)
    buf = conv3d_133(v, sd["conv_init0.0.weight"])
    t = buf
    for i in (0, 2, 4):
        t = leaky_relu(conv3d_133(t, sd[f"conv_init.{i}.weight"]), 0.2)
    buf = t + buf
    if taps is not None:
        taps["init"] = buf
    t = buf
    nblk = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("altblock."))
    for i in range(nblk):
        t = epit_alt_filter(t, sd, f"altblock.{i}.", A)
        if taps is not None and i == 0:
            taps["alt0"] = t
    buf = t + buf
    mosaic = buf.reshape(B, -1, A, A, h, w).transpose(0, 1, 2, 4, 3, 5).reshape(B, -1, A * h, A * w)
    up = pixel_shuffle(conv2d(mosaic, sd["upsampling.0.weight"]), s)
    up = conv2d(leaky_relu(up, 0.2), sd["upsampling.3.weight"], padding=(1, 1))
    return up + sr


# ----------------------------------------------------------------------------------------------
# LFT (model/SR/LFT.py)
# ----------------------------------------------------------------------------------------------


def lft_position_encoding(lengths, token_dim, dtype, temperature=10000):
    """``PositionEncoding.forward`` LFT.py:106-130 for the listed axis lengths: returns one (len, token_dim) table per axis
    (sin of the even columns then cos of the odd columns, concatenated -- not interleaved)."""
    grid = np.arange(token_dim, dtype=np.float64)
    grid = 2 * (grid // 2) / token_dim
    grid = temperature ** grid
    out = []
    for n in lengths:
        pos = np.arange(n, dtype=np.float64).reshape(-1, 1) / grid
        out.append(np.concatenate([np.sin(pos[:, 0::2]), np.cos(pos[:, 1::2])], axis=1).astype(dtype))
    return out


def lft_gen_mask(h, w, k, dtype):
    """``SpaTrans.gen_mask`` LFT.py:161-174 -- the column window is clamped with h, not w (kept)."""
    kl = k // 2
    kr = k - kl
    m = np.full((h, w, h, w), -np.inf, dtype=dtype)
    for i in range(h):
        for j in range(w):
            m[i, j, max(0, i - kl):min(h, i + kr), max(0, j - kl):min(h, j + kr)] = 0.0
    return m.reshape(h * w, h * w)


def _mha_chunked(q, k, v, in_w, out_w, nheads, mask, chunk=8):
    """mha() over the batch axis in chunks (the spatial transformer has L=1024: bound the (N*heads, L, L) scores)."""
    outs = [mha(q[:, i:i + chunk], k[:, i:i + chunk], v[:, i:i + chunk], in_w, out_w, nheads, mask) for i in range(0, q.shape[1], chunk)]
    return np.concatenate(outs, axis=1)


def lft_ang_trans(buf, sd, pre, ang_pe):
    """``AngTrans.forward`` LFT.py:233-246.  buf (b,c,a,h,w)."""
    b, c, a, h, w = buf.shape
    tok = buf.transpose(2, 0, 3, 4, 1).reshape(a, b * h * w, c)                    # a (b h w) c
    tn = layer_norm(tok + ang_pe.reshape(a, 1, c), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    tok = _mha_chunked(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], 8, None, chunk=4096) + tok
    ff = layer_norm(tok, sd[pre + "feed_forward.0.weight"], sd[pre + "feed_forward.0.bias"])
    ff = np.maximum(linear(ff, sd[pre + "feed_forward.1.weight"]), 0)
    tok = linear(ff, sd[pre + "feed_forward.4.weight"]) + tok
    return tok.reshape(a, b, h, w, c).transpose(1, 4, 0, 2, 3)


def lft_spa_trans(buf, sd, pre, spa_pos):
    """``SpaTrans.forward`` LFT.py:188-203.  buf (b,c,a,h,w); spa_pos (1,c,1,h,w)."""
    b, c, a, h, w = buf.shape
    mask = lft_gen_mask(h, w, 5, buf.dtype)
    wm = sd[pre + "MLP.weight"].reshape(-1, c, 3, 3)        # unfold(k3,pad1) + Linear(576->128) == 3x3 conv, LFT.py:176-182

    def sai2token(t):
        n = t.shape[0] * t.shape[2]
        y = conv2d(t.transpose(0, 2, 1, 3, 4).reshape(n, c, h, w), wm, padding=(1, 1))    # (n, 128, h, w)
        return y.reshape(n, -1, h * w).transpose(2, 0, 1)                                 # (h w) n 128
    tok = sai2token(buf)
    pe = sai2token(spa_pos)
    tn = layer_norm(tok + pe, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    tok = _mha_chunked(tn, tn, tok, sd[pre + "attention.in_proj_weight"], sd[pre + "attention.out_proj.weight"], 8, mask, chunk=2) + tok
    ff = layer_norm(tok, sd[pre + "feed_forward.0.weight"], sd[pre + "feed_forward.0.bias"])
    ff = np.maximum(linear(ff, sd[pre + "feed_forward.1.weight"]), 0)
    tok = linear(ff, sd[pre + "feed_forward.4.weight"]) + tok
    t = tok.reshape(h, w, b, a, -1).transpose(2, 4, 3, 0, 1)                               # b c a h w
    wl = sd[pre + "linear.0.weight"].reshape(c, -1)                                        # Conv3d 1x1x1 128->64
    return np.einsum("oc,bcahw->boahw", wl, t)


def lft_forward(x, sd, A, s, dtype=np.float64, taps=None):
    """``get_model.forward`` LFT.py:67-98.  x (B,1,A*h,A*w) -> (B,1,A*h*s,A*w*s)."""
    sd = _cast(sd, dtype)
    x = np.asarray(x, dtype=dtype)
    B, _, Hh, Ww = x.shape
    h, w = Hh // A, Ww // A
    lr = x.reshape(B, 1, A, h, A, w).transpose(0, 1, 2, 4, 3, 5)                           # b c a1 a2 h w
    up = interp_bicubic(lr.reshape(B * A * A, 1, h, w), s).reshape(B, 1, A, A, h * s, w * s)
    up = up.transpose(0, 1, 2, 4, 3, 5).reshape(B, 1, A * h * s, A * w * s)                # LFT.py:263-273
    v = lr.reshape(B, 1, A * A, h, w)
    buf = conv3d_133(v, sd["conv_init0.0.weight"])
    t = buf
    for i in (0, 2, 4):
        t = leaky_relu(conv3d_133(t, sd[f"conv_init.{i}.weight"]), 0.2)
    buf = t + buf
    c = buf.shape[1]
    ph, pw, pa = lft_position_encoding([h, w, A * A], c, dtype)
    spa_pos = ((ph[:, None, :] + pw[None, :, :]) / 2).transpose(2, 0, 1).reshape(1, c, 1, h, w)   # dims [3,4], / len(dim)
    ang_pe = pa                                                                                   # dims [2]
    t = buf
    nblk = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("altblock."))
    for i in range(nblk):
        t = lft_ang_trans(t, sd, f"altblock.{i}.ang_trans.", ang_pe)
        if taps is not None and i == 0:
            taps["ang0"] = t
        t = lft_spa_trans(t, sd, f"altblock.{i}.spa_trans.", spa_pos)
        if taps is not None and i == 0:
            taps["spa0"] = t
    buf = t + buf
    mosaic = buf.reshape(B, c, A, A, h, w).transpose(0, 1, 2, 4, 3, 5).reshape(B, c, A * h, A * w)
    o = pixel_shuffle(conv2d(mosaic, sd["upsampling.0.weight"]), s)
    o = conv2d(leaky_relu(o, 0.2), sd["upsampling.3.weight"], padding=(1, 1))
    return o + up


# ----------------------------------------------------------------------------------------------
# LF_InterNet (model/SR/LF_InterNet.py)
# ----------------------------------------------------------------------------------------------


def internet_forward(x, sd, A, s, dtype=np.float64, n_groups=4, n_layers=4):
    """``get_model.forward`` LF_InterNet.py:33-41 (``make_chains`` :44-67, ``BottleNeck`` :107-124, ``ReconBlock`` :127-141).
    x (B,1,A*h,A*w) -> (B,1,A*h*s,A*w*s).  No global skip."""
    sd = _cast(sd, dtype)
    x = np.asarray(x, dtype=dtype)
    d = (A, A)
    relu = lambda t: np.maximum(t, 0)
    m = sai2macpi(x, A)
    xa = conv2d(m, sd["AngFE.0.weight"], stride=d)
    xs = conv2d(m, sd["SpaFE.0.weight"], dilation=d, padding=d)
    ba, bs = xa, xs
    outs_a, outs_s = [], []
    for g in range(n_groups):
        for l in range(n_layers):
            p = f"CascadeInterBlock.body.{g}.chained_layers.{l}."
            ang2 = relu(conv2d(bs, sd[p + "Spa2Ang.weight"], stride=d))
            spa2 = pixel_shuffle(conv2d(ba, sd[p + "Ang2Spa.0.weight"]), A)
            oa = relu(conv2d(np.concatenate((ba, ang2), 1), sd[p + "AngConvSq.weight"])) + ba
            os_ = relu(conv2d(np.concatenate((bs, spa2), 1), sd[p + "SpaConvSq.weight"], dilation=d, padding=d)) + bs
            ba, bs = oa, os_
        outs_a.append(ba)
        outs_s.append(bs)
    ca, cs = np.concatenate(outs_a, 1), np.concatenate(outs_s, 1)
    a = relu(conv2d(ca, sd["BottleNeck.AngBottle.weight"]))
    cs = np.concatenate((cs, pixel_shuffle(conv2d(a, sd["BottleNeck.Ang2Spa.0.weight"]), A)), 1)
    out = relu(conv2d(cs, sd["BottleNeck.SpaBottle.weight"], dilation=d, padding=d)) + xs
    pre = conv2d(out, sd["ReconBlock.PreConv.weight"], dilation=d, padding=d)
    hr = pixel_shuffle(macpi2sai(pre, A), s)
    return conv2d(hr, sd["ReconBlock.FinalConv.weight"])


# ----------------------------------------------------------------------------------------------
# N2: per-view PSNR / SSIM of cal_metrics (utils/utils.py:91-134; skimage semantics restated -- skimage itself is absent)
# ----------------------------------------------------------------------------------------------


def _gauss11(img):
    """scipy.ndimage.gaussian_filter(img, sigma=1.5, truncate=3.5, mode='reflect') -- what
    skimage.metrics.structural_similarity(gaussian_weights=True) calls: separable 11-tap, edge-including mirror."""
    k = np.exp(-0.5 * (np.arange(-5, 6) / 1.5) ** 2)
    k /= k.sum()
    H, W = img.shape
    ys, xs = _sym(np.arange(-5, H + 5), H), _sym(np.arange(-5, W + 5), W)
    p = img[:, xs]
    h = sum(k[i] * p[:, i:i + W] for i in range(11))
    p = h[ys, :]
    return sum(k[i] * p[i:i + H, :] for i in range(11))


def view_psnr_ssim(label, out, A):
    """label/out (B,1,A*H,A*W) -> (psnr, ssim) arrays (B,A,A), float64: the per-view values cal_metrics averages."""
    label, out = np.asarray(label, np.float64), np.asarray(out, np.float64)
    B, _, Hh, Ww = label.shape
    H, W = Hh // A, Ww // A
    ps, ss = np.zeros((B, A, A)), np.zeros((B, A, A))
    C1, C2, cov = 0.01 ** 2, 0.03 ** 2, 121.0 / 120.0
    for b in range(B):
        for u in range(A):
            for v in range(A):
                x, y = label[b, 0, u * H:(u + 1) * H, v * W:(v + 1) * W], out[b, 0, u * H:(u + 1) * H, v * W:(v + 1) * W]
                mse = np.mean((x - y) ** 2)
                ps[b, u, v] = 10 * np.log10(1.0 / mse) if mse > 0 else np.inf
                ux, uy, uxx, uyy, uxy = _gauss11(x), _gauss11(y), _gauss11(x * x), _gauss11(y * y), _gauss11(x * y)
                vx, vy, vxy = cov * (uxx - ux * ux), cov * (uyy - uy * uy), cov * (uxy - ux * uy)
                S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
                ss[b, u, v] = S[5:-5, 5:-5].mean()
    return ps, ss

# ---- Winograd F(2x2, 3x3) restatement (checks the packed layout and the algebra of csrc/conv3x3_wino.hip) --------------------
# Lavin & Gray's minimal filtering form of the per-view 3x3 correlation of model/SR/DistgSSR.py:22,47,64,79-83,101:
#   Y = At [ sum_c (G g Gt) . (Bt d B) ] A   per 2x2 output tile, d = its 4x4 input patch (zero padded)
WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
WINO_BT = np.array([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
WINO_AT = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def winograd_weights(w):
    """(O, C, 3, 3) -> U[4, 4, O, C] = G g Gt per (o, c), in the dtype of w"""
    return np.einsum("ai,ocij,bj->aboc", WINO_G.astype(w.dtype), w, WINO_G.astype(w.dtype))


def winograd_pack(w):
    """the device layout of lfsr_pack_conv_weight's Winograd part for a (64, 64, 3, 3) weight:
    [j = k/8][nt = n/32][p = 4 xi + nu][half = (k/4)&1][n%32][k%4], flattened (fp64 compute, one rounding to fp32)"""
    U = winograd_weights(w.astype(np.float64)).reshape(16, 64, 64)          # [p][n][k]
    U = U.reshape(16, 2, 32, 8, 2, 4)                                       # p, nt, n32, j, half, e
    return np.ascontiguousarray(U.transpose(3, 1, 0, 4, 2, 5)).astype(np.float32).reshape(-1)


def conv3x3_winograd(x, w):
    """per-image 3x3 correlation, zero pad 1, via F(2x2,3x3): x (N, C, H, W) with even H, W; w (O, C, 3, 3)"""
    N, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    U = winograd_weights(w)
    y = np.zeros((N, w.shape[0], H, W), dtype=x.dtype)
    BT, AT = WINO_BT.astype(x.dtype), WINO_AT.astype(x.dtype)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = np.einsum("ai,ncij,bj->abnc", BT, d, BT)
            M = np.einsum("aboc,abnc->abno", U, V)
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ia,abno,jb->noij", AT, M, AT)
    return y


# F(4x4,3x3): the form the default HIP 3x3 kernel runs (conv3x3_wino4.hip); interpolation points 0, +-1, +-2, inf
WINO4_BT = np.array([[4.0, 0.0, -5.0, 0.0, 1.0, 0.0], [0.0, -4.0, -4.0, 1.0, 1.0, 0.0], [0.0, 4.0, -4.0, -1.0, 1.0, 0.0],
                     [0.0, -2.0, -1.0, 2.0, 1.0, 0.0], [0.0, 2.0, -1.0, -2.0, 1.0, 0.0], [0.0, 4.0, 0.0, -5.0, 0.0, 1.0]])
WINO4_G = np.array([[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                    [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]])
WINO4_AT = np.array([[1.0, 1.0, 1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, 2.0, -2.0, 0.0],
                     [0.0, 1.0, 1.0, 4.0, 4.0, 0.0], [0.0, 1.0, -1.0, 8.0, -8.0, 1.0]])


def winograd4_weights(w):
    """(O, C, 3, 3) -> U[6, 6, O, C] = G g Gt per (o, c), in the dtype of w"""
    return np.einsum("ai,ocij,bj->aboc", WINO4_G.astype(w.dtype), w, WINO4_G.astype(w.dtype))


def winograd4_pack(w):
    """the device layout of the F(4x4,3x3) part of lfsr_pack_conv_weight for a (64, 64, 3, 3) weight:
    [s = k/4][ns = n/16][q = p/4][lane = 16 (k%4) + n%16][e = p%4], p = 18 (nu // 3) + 3 xi + nu % 3 -- nu-half major, the order in which the
    kernel keeps the 36 transform positions (a half tile's wave owns 18 contiguous ones) (fp64 compute, one rounding to fp32)"""
    order = [6 * xi + 3 * hf + n for hf in range(2) for xi in range(6) for n in range(3)]        # natural index 6 xi + nu of place p
    U = winograd4_weights(w.astype(np.float64)).reshape(36, 64, 64)[order].reshape(9, 4, 4, 16, 16, 4)   # q, e, ns, m, s, kq
    return np.ascontiguousarray(U.transpose(4, 2, 0, 5, 3, 1)).astype(np.float32).reshape(-1)


def conv3x3_winograd4(x, w):
    """per-image 3x3 correlation, zero pad 1, via F(4x4,3x3): x (N, C, H, W) with H, W multiples of 4; w (O, C, 3, 3)"""
    N, C, H, W = x.shape
    assert H % 4 == 0 and W % 4 == 0
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    U = winograd4_weights(w)
    y = np.zeros((N, w.shape[0], H, W), dtype=x.dtype)
    BT, AT = WINO4_BT.astype(x.dtype), WINO4_AT.astype(x.dtype)
    for ty in range(H // 4):
        for tx in range(W // 4):
            d = xp[:, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]
            V = np.einsum("ai,ncij,bj->abnc", BT, d, BT)
            M = np.einsum("aboc,abnc->abno", U, V)
            y[:, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = np.einsum("ia,abno,jb->noij", AT, M, AT)
    return y


# ---- N4 output tail (utils/utils.py:191-204 ycbcr2rgb; train.py:329-341) ------------------------------------------------------
def ycbcr2rgb(x):
    """(H, W, 3) YCbCr in [0, 1] -> RGB (float64): the reference's formula, operation for operation"""
    mat = np.array([[65.481, 128.553, 24.966], [-37.797, -74.203, 112.0], [112.0, -93.786, -18.214]])
    mat_inv = np.linalg.inv(mat)
    offset = np.matmul(mat_inv, np.array([16, 128, 128]))
    mat_inv = mat_inv * 255
    y = np.zeros(x.shape, dtype="double")
    for i in range(3):
        y[:, :, i] = mat_inv[i, 0] * x[:, :, 0] + mat_inv[i, 1] * x[:, :, 1] + mat_inv[i, 2] * x[:, :, 2] - offset[i]
    return y


def sr_views_rgb_u8(sr_y, sr_cbcr, A):
    """(A h, A w) Y and (2, A h, A w) CbCr -> (A, A, h, w, 3) uint8, as train.py:332-334 builds Sr_4D_rgb"""
    ycc = np.concatenate([sr_y[None], sr_cbcr], axis=0).transpose(1, 2, 0)
    rgb = (ycbcr2rgb(ycc).clip(0, 1) * 255).astype("uint8")
    H, W, _ = rgb.shape
    return rgb.reshape(A, H // A, A, W // A, 3).transpose(0, 2, 1, 3, 4)
