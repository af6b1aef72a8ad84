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


def gen_aux(out):
    """Pins for the "next" rows N3 / N4 taken from the reference itself:
    * utils/masked_pretraining.py (torch only -> imported by file path): for every strategy and mask_value in {zero, mean} the
      sequence of (masked?, mask_indices) of 8 training-mode calls under ``random.seed(4)`` and the masked tensors;
    * utils/utils.py:181-204 ``rgb2ycbcr`` / ``ycbcr2rgb`` (AST-extracted: the module itself needs skimage/xlwt) driven exactly as
      train.py:332-334 drives them: uint8 view stacks.
    N2 (SSIM): skimage is not installed in this image, so the reference's cal_metrics cannot be run; the oracle's SSIM restatement stays
    pinned on scipy.ndimage.gaussian_filter only (tests/test_metrics_masking.py) -- recorded in aux.json."""
    import importlib.util
    import random
    from einops import rearrange
    spec = importlib.util.spec_from_file_location("ref_masked_pretraining", os.path.join(REF, "utils/masked_pretraining.py"))
    MP = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MP)
    A, h, w = 5, 6, 8
    x = torch.from_numpy(synth_input((3, 1, A * h, A * w), seed=5))
    meta = {"masking": {"A": A, "h": h, "w": w, "B": 3, "input_seed": 5, "python_random_seed": 4, "mask_ratio": 0.3, "calls": 8, "cases": {}},
            "ssim_pin": "scipy.ndimage.gaussian_filter only (skimage absent from the image: utils/utils.py:91-134 cannot be imported)"}
    arrs = {}
    for strategy in ("random", "grid", "corners", "center"):
        for value in ("zero", "mean"):
            m = MP.MaskedAngularPretraining(angRes=A, mask_ratio=0.3, mask_strategy=strategy, mask_value=value).train()
            random.seed(4)
            calls = []
            for c in range(8):
                y, info = m(x)
                if info["masked"]:
                    calls.append([[int(i), int(j)] for (i, j) in info["mask_indices"]])
                    if value == "mean":
                        arrs[f"mask_{strategy}_{value}_{c}"] = y.numpy()
                    else:   # 'zero' output follows from the indices: keep its checksum only
                        meta["masking"].setdefault("zero_sha256", {})[f"{strategy}_{c}"] = sha(y.numpy())
                else:
                    assert y is x
                    calls.append(None)
            meta["masking"]["cases"][f"{strategy}_{value}"] = calls
    pm = MP.ProgressiveMasking(angRes=A, start_ratio=0.1, end_ratio=0.4, warmup_epochs=20)
    prog = []
    for ep in (0, 5, 10, 20, 30):
        pm.set_epoch(ep)
        prog.append([ep, pm.masker.mask_ratio, pm.masker.num_masked])
    meta["masking"]["progressive"] = prog
    # ---- YCbCr <-> RGB (utils/utils.py:181-204), driven as train.py:332-334 does
    src = open(os.path.join(REF, "utils/utils.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("rgb2ycbcr", "ycbcr2rgb")]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "ref_color", "exec"), ns)
    rng = np.random.default_rng(8)
    ycc_cases = {}
    for (A2, h2, w2) in [(5, 32, 32), (3, 7, 10)]:
        y = (rng.random((A2 * h2, A2 * w2)) * 1.2 - 0.1).astype(np.float32)
        cc = (rng.random((2, A2 * h2, A2 * w2)) * 1.2 - 0.1).astype(np.float32)
        Sr_SAI_ycbcr = torch.cat((torch.from_numpy(y)[None, None], torch.from_numpy(cc)[None]), dim=1)
        Sr_SAI_rgb = (ns["ycbcr2rgb"](Sr_SAI_ycbcr.squeeze().permute(1, 2, 0).numpy()).clip(0, 1) * 255).astype("uint8")
        Sr_4D_rgb = rearrange(Sr_SAI_rgb, "(a1 h) (a2 w) c -> a1 a2 h w c", a1=A2, a2=A2)
        key = f"A{A2}_{h2}x{w2}"
        arrs["ycc_y_" + key], arrs["ycc_cbcr_" + key], arrs["rgb_u8_" + key] = y, cc, Sr_4D_rgb
        ycc_cases[key] = dict(A=A2, h=h2, w=w2)
    rgb = rng.random((9, 13, 3))
    arrs["rgb_in"], arrs["ycbcr_of_rgb"] = rgb, ns["rgb2ycbcr"](rgb)
    meta["ycbcr"] = {"cases": ycc_cases, "numpy": np.__version__}
    np.savez_compressed(os.path.join(out, "aux.npz"), **arrs)
    json.dump(meta, open(os.path.join(out, "aux.json"), "w"), indent=1)
    print("aux:", len(arrs), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "aux":
        gen_aux(HERE)
        sys.exit(0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen_index_ops(HERE)
    gen_models(HERE)
    gen_distg_grads(HERE)
    gen_aux(HERE)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("fixture bytes:", tot)
