import json
import os

import numpy as np

from lfsr_amd.synth import synth_input, synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def models_meta():
    return json.load(open(os.path.join(GOLDEN, "models.json")))


def model_case(name, tag):
    """-> (case dict, state_dict (numpy fp32), input (numpy fp32), golden arrays npz)"""
    meta = models_meta()["models"][name]
    case = meta["full"] if tag == "full" else meta["cases"][tag]
    sd = synth_state_dict([(k, tuple(s)) for k, s in case["spec"]], seed=0)
    x = synth_input((case["B"], 1, case["A"] * case["h"], case["A"] * case["w"]), seed=1)
    npz = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    return case, sd, x, npz


def psnr(a, b):
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


# ---------------------------------------------------------------------------------------------------------------------
# LeakyReLU' masks of the DistgSSR training forward: the HIP path's saved activations (lfsr_distgssr_train_saved) <-> the layouts of the
# reference graph (oracle/lfsr_torch_port.py:distg_block's rec / force dictionaries)
# ---------------------------------------------------------------------------------------------------------------------
MASK_KINDS = ("S1", "S2", "A1", "A2", "EH1", "EH2", "EV1", "EV2", "FZ")
_SAVED = {"S1": (0, None), "S2": (1, (0, 64)), "A1": (2, None), "A2": (1, (64, 80)), "EH1": (3, None), "EH2": (1, (80, 112)),
          "EV1": (4, None), "EV2": (1, (112, 144)), "FZ": (5, None)}      # kind -> (which of lfsr_distgssr_train_saved, channel slice of the concat buffer)
_PERMS = {}


def mask_ref_shape(kind, B, A, h, w):
    return {"S1": (B, 64, h * A, w * A), "S2": (B, 64, h * A, w * A), "FZ": (B, 64, h * A, w * A), "A1": (B, 16, h, w), "A2": (B, 16 * A * A, h, w),
            "EH1": (B, 32, h * A, w), "EH2": (B, 32 * A, h * A, w), "EV1": (B, 32, w * A, h), "EV2": (B, 32 * A, w * A, h)}[kind]


def _ref_to_hip(kind, t, B, A, h, w):
    """a tensor in the reference layout of `kind` -> flat, in the order the HIP path stores that activation"""
    import torch
    import torch.nn.functional as F

    def vcl(z):   # NCHW MacPI (B,C,h*A,w*A) -> [b][u][v][y][x][c]
        return z.reshape(B, z.shape[1], h, A, w, A).permute(0, 3, 5, 2, 4, 1).reshape(-1)

    def ps1d(z, f):   # DistgSSR.py:114-131 (factor-major channel order)
        Bz, fC, Hh, Ww = z.shape
        return z.reshape(Bz, f, fC // f, Hh, Ww).permute(0, 2, 3, 4, 1).reshape(Bz, fC // f, Hh, Ww * f)
    if kind in ("S1", "S2", "FZ"):
        return vcl(t)
    if kind == "A1":
        return t.permute(0, 2, 3, 1).reshape(-1)
    if kind == "A2":
        return vcl(F.pixel_shuffle(t, A))
    if kind == "EH1":
        return t.reshape(B, 32, h, A, w).permute(0, 3, 2, 4, 1).reshape(-1)
    if kind == "EV1":
        return t.reshape(B, 32, w, A, h).permute(0, 3, 4, 2, 1).reshape(-1)
    if kind == "EH2":
        return vcl(ps1d(t, A))
    if kind == "EV2":
        return vcl(ps1d(t, A).permute(0, 1, 3, 2))
    raise KeyError(kind)


def mask_perm(kind, B, A, h, w):
    """perm with hip_flat[i] = ref_flat[perm[i]] (an index tensor pushed through the reference -> HIP layout map; cached)"""
    import torch
    key = (kind, B, A, h, w)
    if key not in _PERMS:
        shp = mask_ref_shape(kind, B, A, h, w)
        n = int(np.prod(shp))
        _PERMS[key] = _ref_to_hip(kind, torch.arange(n, dtype=torch.float64).reshape(shp), B, A, h, w).long()
    return _PERMS[key]


def hip_saved_mask(rt, x, kind, index):
    """LeakyReLU output signs (> 0) of block `index` as the HIP training forward saved them, flat in HIP order (CPU bool tensor)"""
    which, sl = _SAVED[kind]
    v = rt.train_saved(x, which, index)
    if sl is not None:
        v = v.reshape(-1, 144)[:, sl[0]:sl[1]]
    return (v > 0).reshape(-1).cpu()


def hip_mask_to_ref(mask_flat, kind, B, A, h, w):
    """the same mask in the reference graph's layout (what distg_block's `force` consumes)"""
    import torch
    shp = mask_ref_shape(kind, B, A, h, w)
    out = torch.empty(int(np.prod(shp)), dtype=torch.bool)
    out[mask_perm(kind, B, A, h, w)] = mask_flat
    return out.reshape(shp)


def ref_mask_to_hip(mask_ref, kind, B, A, h, w):
    return mask_ref.reshape(-1)[mask_perm(kind, B, A, h, w)]
