"""Hot-path half of the reference's ``utils/utils.py``: ``ImageExtend`` / ``LFdivide`` / ``LFintegrate``
(utils/utils.py:137-178) with the same names, argument order and meaning, running on the MI355X through the
C ABI.  The metric / logging / colour helpers of that file are host code outside this path (SURVEY section 2
rows 8, 11) and are not mirrored; importing this module does NOT parse ``sys.argv`` (the reference's does,
utils/utils.py:8 -> option.py:36)."""
from lfsr_amd import capi


def ImageExtend(Im, bdr):
    return capi.image_extend(Im.contiguous(), list(bdr))


def LFdivide(data, angRes, patch_size, stride):
    return capi.lf_divide(data.contiguous(), angRes, patch_size, stride)


def LFintegrate(subLF, angRes, pz, stride, h, w):
    return capi.lf_integrate(subLF.contiguous(), angRes, pz, stride, h, w)


# ---- N4: the output tail of test() (reference utils/utils.py:181-204, train.py:329-341, inference.py:205-216) -----------------
_YCBCR_M = ((65.481, 128.553, 24.966), (-37.797, -74.203, 112.0), (112.0, -93.786, -18.214))


def _ycbcr_inverse():
    """inv(M) * 255 and inv(M) . (16, 128, 128), computed the way the reference's ycbcr2rgb does (numpy float64)"""
    import numpy as np
    mat = np.array(_YCBCR_M)
    mat_inv = np.linalg.inv(mat)
    offset = np.matmul(mat_inv, np.array([16, 128, 128]))
    return mat_inv * 255, offset


def rgb2ycbcr(x):
    """(H, W, 3) RGB in [0, 1] -> YCbCr in [0, 1] (float64), same formula as the reference (utils/utils.py:181-189); torch or numpy"""
    import numpy as np
    import torch
    t = torch.as_tensor(x).double() if not isinstance(x, np.ndarray) else torch.from_numpy(x).double()
    m = torch.tensor(_YCBCR_M, dtype=torch.float64, device=t.device)
    off = torch.tensor([16.0, 128.0, 128.0], dtype=torch.float64, device=t.device)
    y = torch.stack([m[i, 0] * t[..., 0] + m[i, 1] * t[..., 1] + m[i, 2] * t[..., 2] + off[i] for i in range(3)], dim=-1) / 255.0
    return y.cpu().numpy() if isinstance(x, np.ndarray) else y


def ycbcr2rgb_views(Sr_SAI_y, Sr_SAI_cbcr, angRes):
    """device form of  (ycbcr2rgb(cat(y, cbcr)).clip(0, 1) * 255).astype('uint8')  + the split into views:
    Sr_SAI_y (1, 1, A h, A w) / (A h, A w), Sr_SAI_cbcr (1, 2, A h, A w) / (2, A h, A w) fp32 device tensors -> (A, A, h, w, 3) uint8"""
    import ctypes as C
    import torch
    y = Sr_SAI_y.reshape(Sr_SAI_y.shape[-2], Sr_SAI_y.shape[-1]).contiguous().float()
    cc = Sr_SAI_cbcr.reshape(2, y.shape[0], y.shape[1]).contiguous().float()
    if not y.is_cuda:
        raise capi.LfsrError("ycbcr2rgb_views needs device tensors (there is no CPU fallback for the HIP path)")
    A = int(angRes)
    h, w = y.shape[0] // A, y.shape[1] // A
    out = torch.empty(A, A, h, w, 3, dtype=torch.uint8, device=y.device)
    m255, off = _ycbcr_inverse()
    m_arr = (C.c_double * 9)(*[float(v) for v in m255.reshape(-1)])
    o_arr = (C.c_double * 3)(*[float(v) for v in off])
    capi.check(capi.load().lfsr_ycbcr2rgb_views(capi.dev_ptr(y), capi.dev_ptr(cc), capi.dev_ptr(out), A, h, w, m_arr, o_arr, capi.stream_ptr()),
               "ycbcr2rgb_views")
    return out


def write_bmp(path, img):
    """(h, w, 3) uint8 RGB -> 24-bit uncompressed BMP (bottom-up rows, BGR, rows padded to 4 bytes): what imageio.imwrite(path, img)
    produces for the reference's View_i_j.bmp files (train.py:341), without the imageio dependency"""
    import struct
    import numpy as np
    a = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
    hh, ww, _ = a.shape
    row = (ww * 3 + 3) // 4 * 4
    body = np.zeros((hh, row), dtype=np.uint8)
    body[:, :ww * 3] = a[::-1, :, ::-1].reshape(hh, ww * 3)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + row * hh, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, ww, hh, 1, 24, 0, row * hh, 2835, 2835, 0, 0))
        f.write(body.tobytes())


def save_views_bmp(save_dir, views_u8, prefix="View_"):
    """views_u8 (A, A, h, w, 3) uint8 (device or host) -> <save_dir>/View_i_j.bmp, the CodaBench naming of train.py:336-341"""
    import os
    v = views_u8.cpu().numpy() if hasattr(views_u8, "cpu") else views_u8
    os.makedirs(str(save_dir), exist_ok=True)
    for i in range(v.shape[0]):
        for j in range(v.shape[1]):
            write_bmp(os.path.join(str(save_dir), prefix + str(i) + "_" + str(j) + ".bmp"), v[i, j])
