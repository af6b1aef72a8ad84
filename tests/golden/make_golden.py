#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the *reference* implementation on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's Python files are imported from where they lie (``/root/reference``); nothing of
them is copied here.  ``utils/utils.py`` cannot be imported whole (it needs ``skimage``/``xlwt`` and
parses ``sys.argv`` through ``option.py:36``), so ``ImageExtend``/``LFdivide``/``LFintegrate`` are
taken by executing only those three ``def`` nodes of its AST at generation time.

Weights are NOT stored: they are regenerated from ``lfsr_amd.synth`` (numpy PCG64) on both sides, the
fixture keeps the state_dict (key, shape) contract plus inputs' seeds and the reference outputs.
"""
import ast
import hashlib
import importlib
import json
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from lfsr_amd.synth import synth_input, synth_state_dict  # noqa: E402

sys.path.insert(0, REF)  # reference's ``model.SR.*`` (our mirror lives inside the package dir, not on sys.path)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_utils():
    src = open(os.path.join(REF, "utils/utils.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("ImageExtend", "LFdivide", "LFintegrate")]
    mod = ast.Module(body=keep, type_ignores=[])
    import torch.nn.functional as F
    from einops import rearrange
    ns = {"torch": torch, "F": F, "rearrange": rearrange}
    exec(compile(mod, "ref_utils", "exec"), ns)
    return ns


def load_ref_model(name, A, s, seed=0):
    M = importlib.import_module("model.SR." + name)
    net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=s))
    spec = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synth_state_dict(spec, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return M, net, spec


def gen_index_ops(out):
    D = importlib.import_module("model.SR.DistgSSR")
    meta = {}
    arrs = {}
    # a1/a2 SAI<->MacPI  (DistgSSR.py:134-155)
    for tag, (B, C, A, h, w) in {"s2m_a": (2, 3, 5, 4, 6), "s2m_b": (1, 2, 3, 7, 5)}.items():
        x = torch.arange(B * C * A * h * A * w, dtype=torch.float32).view(B, C, A * h, A * w)
        y = D.SAI2MacPI(x, A)
        z = D.MacPI2SAI(x, A)
        assert torch.equal(D.MacPI2SAI(y, A), x)
        arrs[tag + "_sai2macpi"] = y.numpy().astype(np.int32)
        arrs[tag + "_macpi2sai"] = z.numpy().astype(np.int32)
        meta[tag] = dict(B=B, C=C, A=A, h=h, w=w)
    # a3 PixelShuffle (torch) and a4 PixelShuffle1D (DistgSSR.py:114-131)
    for tag, (B, C, r, h, w) in {"ps_a": (2, 3, 5, 3, 4), "ps_b": (1, 2, 4, 5, 3), "ps_c": (1, 4, 2, 2, 2)}.items():
        x = torch.arange(B * C * r * r * h * w, dtype=torch.float32).view(B, C * r * r, h, w)
        arrs[tag] = torch.nn.PixelShuffle(r)(x).numpy().astype(np.int32)
        meta[tag] = dict(B=B, C=C, r=r, h=h, w=w)
    for tag, (B, C, f, h, w) in {"ps1d_a": (2, 3, 5, 4, 3), "ps1d_b": (1, 2, 3, 2, 5)}.items():
        x = torch.arange(B * C * f * h * w, dtype=torch.float32).view(B, C * f, h, w)
        arrs[tag] = D.PixelShuffle1D(f)(x).numpy().astype(np.int32)
        meta[tag] = dict(B=B, C=C, f=f, h=h, w=w)
    # a5-a7 ImageExtend / LFdivide / LFintegrate (utils/utils.py:137-178)
    U = ref_utils()
    x = torch.arange(3 * 2 * 5 * 7, dtype=torch.float32).view(3, 2, 5, 7)
    arrs["imext"] = U["ImageExtend"](x, [2, 4, 3, 6]).numpy().astype(np.int32)
    meta["imext"] = dict(shape=[3, 2, 5, 7], bdr=[2, 4, 3, 6])
    div = {}
    for (A, h0, w0) in [(5, 128, 128), (5, 125, 125), (5, 108, 156), (5, 40, 33), (3, 37, 50)]:
        P, S = 32, 16
        x = torch.arange(A * h0 * A * w0, dtype=torch.float32).view(A * h0, A * w0)
        sub = U["LFdivide"](x, A, P, S)
        back = U["LFintegrate"](sub, A, P, S, h0, w0)
        assert torch.equal(back.permute(0, 2, 1, 3).reshape(A * h0, A * w0), x), "round trip"
        # x4 'SR' stand-in: nearest x4 of each sub-patch so LFintegrate at scale 4 has a known answer
        s = 4
        big = torch.from_numpy((np.arange(sub.numel() * s * s, dtype=np.int64) % 16777213).astype(np.float32)).view(
            sub.shape[0], sub.shape[1], sub.shape[2] * s, sub.shape[3] * s)
        integ = U["LFintegrate"](big, A, P * s, S * s, h0 * s, w0 * s)
        key = f"A{A}_{h0}x{w0}"
        div[key] = dict(A=A, h0=h0, w0=w0, P=P, S=S, numU=int(sub.shape[0]), numV=int(sub.shape[1]),
                        divide_sha=sha(sub.numpy().astype(np.int32)),
                        integrate_s4_shape=list(integ.shape), integrate_s4_sha=sha(integ.numpy().astype(np.int32)))
        if (h0, w0) in [(40, 33), (37, 50)]:
            arrs["div_" + key] = sub.numpy().astype(np.int32)
    meta["lfdivide"] = div
    np.savez_compressed(os.path.join(out, "index_ops.npz"), **arrs)
    json.dump(meta, open(os.path.join(out, "index_ops.json"), "w"), indent=1)
    print("index ops:", {k: v.shape for k, v in arrs.items()})


MODEL_CASES = {
    # name: list of (tag, A, h, w, s, B)
    "DistgSSR": [("a5h8s4", 5, 8, 8, 4, 1), ("a3h6w8s2", 3, 6, 8, 2, 2)],
    "EPIT": [("a5h8s4", 5, 8, 8, 4, 1), ("a3h6w8s2", 3, 6, 8, 2, 2)],
    "LFT": [("a5h8s4", 5, 8, 8, 4, 1), ("a3h6w8s2", 3, 6, 8, 2, 2)],
    "LF_InterNet": [("a5h8s2", 5, 8, 8, 2, 1), ("a3h6w8s4", 3, 6, 8, 4, 2)],
}
FULL_CASES = {"DistgSSR": (5, 32, 32, 4, 1), "EPIT": (5, 32, 32, 4, 1), "LFT": (5, 32, 32, 4, 1), "LF_InterNet": (5, 32, 32, 2, 1)}

DISTG_TAPS = {  # intermediates of the first DistgSSR case (module name -> fixture key)
    "init_conv": "init_conv",
    "disentg.Group.0.Block.0.SpaConv": "b0_spa",
    "disentg.Group.0.Block.0.AngConv": "b0_ang",
    "disentg.Group.0.Block.0.EPIConv": "b0_epi_last",   # called twice (H then V^T): hook keeps both
    "disentg.Group.0.Block.0": "b0_out",
    "disentg.Group.0": "g0_out",
    "disentg": "disentg_out",
}


def gen_models(out):
    meta = {"torch": torch.__version__, "numpy": np.__version__, "weights_seed": 0, "input_seed": 1, "models": {}}
    for name, cases in MODEL_CASES.items():
        mm = {"cases": {}}
        arrs = {}
        for (tag, A, h, w, s, B) in cases:
            M, net, spec = load_ref_model(name, A, s)
            x = torch.from_numpy(synth_input((B, 1, A * h, A * w), seed=1))
            caps = {}
            hooks = []
            if name == "DistgSSR" and tag == "a5h8s4":
                mods = dict(net.named_modules())
                for mn, key in DISTG_TAPS.items():
                    def mk(key):
                        def f(m, i, o):
                            caps.setdefault(key, []).append(o.detach().numpy().copy())
                        return f
                    hooks.append(mods[mn].register_forward_hook(mk(key)))
            with torch.no_grad():
                y = net(x, [A, A])
            for hk in hooks:
                hk.remove()
            arrs[f"{tag}_out"] = y.numpy()
            for key, lst in caps.items():
                for i, a in enumerate(lst):
                    arrs[f"{tag}_{key}_{i}"] = a
            mm["cases"][tag] = dict(A=A, h=h, w=w, s=s, B=B, spec=[[k, list(sh)] for k, sh in spec],
                                    out_shape=list(y.shape), n_params=int(sum(int(np.prod(sh)) for _, sh in spec)))
            print(name, tag, tuple(y.shape), float(y.mean()), float(y.std()))
        # full-size case: checksum + strided sample
        A, h, w, s, B = FULL_CASES[name]
        M, net, spec = load_ref_model(name, A, s)
        x = torch.from_numpy(synth_input((B, 1, A * h, A * w), seed=1))
        with torch.no_grad():
            y = net(x, [A, A]).numpy()
        arrs["full_sample"] = y[:, :, ::8, ::8].copy()
        mm["full"] = dict(A=A, h=h, w=w, s=s, B=B, spec=[[k, list(sh)] for k, sh in spec], out_shape=list(y.shape),
                          mean=float(y.mean()), std=float(y.std()), sha256_f32=sha(y),
                          n_params=int(sum(int(np.prod(sh)) for _, sh in spec)))
        print(name, "full", y.shape, mm["full"]["mean"], mm["full"]["std"], mm["full"]["n_params"])
        np.savez_compressed(os.path.join(out, f"model_{name}.npz"), **arrs)
        meta["models"][name] = mm
    json.dump(meta, open(os.path.join(out, "models.json"), "w"), indent=1)


def gen_distg_grads(out):
    """DistgSSR fwd+bwd with get_loss (L1) -- train.py:256-264 without AMP (fp32)."""
    A, h, w, s, B = 3, 6, 8, 2, 2
    M, net, spec = load_ref_model("DistgSSR", A, s)
    net.train()
    x = torch.from_numpy(synth_input((B, 1, A * h, A * w), seed=1))
    label = torch.from_numpy(synth_input((B, 1, A * h * s, A * w * s), seed=2))
    crit = M.get_loss(None)
    out_t = net(x, [A, A])
    loss = crit(out_t, label, [A, A])
    loss.backward()
    arrs = {"loss": np.float64(loss.item())}
    names, norms, projs = [], [], []
    for k, p in net.named_parameters():
        g = p.grad.detach().numpy().astype(np.float64)
        probe = np.random.default_rng([7, len(names)]).standard_normal(g.shape)
        names.append(k)
        norms.append(np.sqrt((g * g).sum()))
        projs.append((g * probe).sum())
        if g.size <= 2400:
            arrs["grad::" + k] = g.astype(np.float32)
    arrs["norms"] = np.array(norms)
    arrs["projs"] = np.array(projs)
    np.savez_compressed(os.path.join(out, "distg_grads.npz"), **arrs)
    json.dump(dict(A=A, h=h, w=w, s=s, B=B, names=names, label_seed=2), open(os.path.join(out, "distg_grads.json"), "w"), indent=1)
    print("grads: loss", loss.item(), "n", len(names))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen_index_ops(HERE)
    gen_models(HERE)
    gen_distg_grads(HERE)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("fixture bytes:", tot)
