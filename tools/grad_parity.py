#!/usr/bin/env python3
"""Attribution of the BASELINE-geometry gradient error (reference: train.py:256-264).

(A,h,w,s,B) = (5,32,32,4,2), every one of the 137 parameter gradients of DistgSSR, L1 loss.  Truth = fp64 autograd over the stock-torch CPU form of the
oracle; yardstick = the SAME graph in fp32 (what the reference computes on the CPU).  The HIP backward is run once per kernel selection
(forward 3x3 form / data-gradient 3x3 form / weight-gradient 3x3 form) and, per selection, the tool records
  * per parameter e_hip = ||g_hip - g_64|| / ||g_64|| and e_ref = ||g_32 - g_64|| / ||g_64||, their distributions and the worst list,
  * how many tensors break "e_hip <= max(1e-4, 3 e_ref)",
  * LeakyReLU' mask flips: elements whose saved post-activation sign differs from the fp64 graph's, per layer kind (the fp32 CPU graph's flips beside them),
  * (first three selections) the DECOMPOSITION: the fp64 graph re-run with the HIP forward's own LeakyReLU decisions forced (oracle `force=`) gives
    g_64|hip-masks; e_arith = ||g_hip - g_64|hip-masks|| / ||g_64|| is the arithmetic error alone, e_flip = ||g_64|hip-masks - g_64|| / ||g_64|| what the
    handful of flipped decisions moves.
Writes profiles/r03_grad_parity.json (argument 1 overrides the path)."""
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("LFSR_LAB", "1")     # this tool drives the library's A/B selectors
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lfsr_amd import capi                                   # noqa: E402
from lfsr_amd.synth import synth_input, synth_state_dict    # noqa: E402
from oracle import lfsr_torch_port as T                     # noqa: E402  (checker)
from tests import helpers as TH                             # noqa: E402

A, h, w, s, B = 5, 32, 32, 4, int(os.environ.get("GP_BATCH", "2"))
MODES = [
    ("default: fwd F(4x4) | dgrad F(4x4) | wgrad F(2x2)", {}, True),
    ("LFSR_CONV3X3=wino2: fwd F(2x2) | dgrad F(2x2) | wgrad F(2x2)", {"LFSR_CONV3X3": "wino2"}, True),
    ("LFSR_CONV3X3=halo LFSR_WGRAD3=direct: all direct", {"LFSR_CONV3X3": "halo", "LFSR_WGRAD3": "direct"}, True),
    ("LFSR_DGRAD3=wino2: fwd F(4x4) | dgrad F(2x2) | wgrad F(2x2)", {"LFSR_DGRAD3": "wino2"}, False),
    ("LFSR_DGRAD3=halo: fwd F(4x4) | dgrad direct | wgrad F(2x2)", {"LFSR_DGRAD3": "halo"}, False),
    ("LFSR_CONV3X3=wino2 LFSR_DGRAD3=wino4: fwd F(2x2) | dgrad F(4x4) | wgrad F(2x2)", {"LFSR_CONV3X3": "wino2", "LFSR_DGRAD3": "wino4"}, False),
    ("LFSR_CONV3X3=halo: fwd direct | dgrad direct | wgrad F(2x2)", {"LFSR_CONV3X3": "halo"}, False),
    ("LFSR_WGRAD3=direct: fwd F(4x4) | dgrad F(4x4) | wgrad direct", {"LFSR_WGRAD3": "direct"}, False),
]
KINDS = ("S1", "S2", "A1", "A2", "EH1", "EH2", "EV1", "EV2", "FZ")


def load_plugin():
    import importlib
    sys.path.insert(0, capi._HERE)
    try:
        return importlib.import_module("model.SR.DistgSSR")
    finally:
        sys.path.remove(capi._HERE)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_grad_parity.json")
    torch.set_num_threads(int(os.environ.get("GP_THREADS", "16")))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(sh)) for k, sh in meta["spec"]], seed=0)
    x = synth_input((B, 1, A * h, A * w), seed=1)
    label = synth_input((B, 1, A * h * s, A * w * s), seed=2)
    blocks = [f"disentg.Group.{g}.Block.{b}." for g in range(4) for b in range(4)]
    grads, masks, losses = {}, {}, {}

    def cpu_graph(dt, force=None):
        t0 = time.time()
        sdt = {k: torch.from_numpy(v).to(dt).requires_grad_(True) for k, v in sd.items()}
        rec = {} if force is None else None
        with torch.enable_grad():
            rl = torch.nn.functional.l1_loss(T.distgssr_forward_graph(torch.from_numpy(x).to(dt), sdt, A, s, rec=rec, force=force), torch.from_numpy(label).to(dt))
        rl.backward()
        print(f"CPU {dt}{' with forced masks' if force is not None else ''}: loss {rl.item():.9f}  ({time.time() - t0:.1f} s)", flush=True)
        return {k: v.grad.double() for k, v in sdt.items()}, rec, rl.item()
    for dt in (torch.float64, torch.float32):
        grads[dt], rec, losses[dt] = cpu_graph(dt)
        masks[dt] = {key: TH.ref_mask_to_hip(m, key.rsplit(".", 1)[1], B, A, h, w) for key, m in rec.items()}      # flat, HIP order
        if dt == torch.float32:
            rec32 = rec
        del rec
    # the yardstick's own decomposition: the fp64 graph with the fp32 CPU graph's LeakyReLU decisions
    g_m32, _, _ = cpu_graph(torch.float64, force=rec32)
    del rec32
    elements = {k: 0 for k in TH.MASK_KINDS}
    ref_flips = {k: 0 for k in TH.MASK_KINDS}
    for pre in blocks:
        for k in TH.MASK_KINDS:
            ref_flips[k] += int((masks[torch.float64][pre + k] != masks[torch.float32][pre + k]).sum())
            elements[k] += masks[torch.float64][pre + k].numel()
    n64 = {k: g.norm().clamp_min(1e-30) for k, g in grads[torch.float64].items()}
    e_ref = {k: float((grads[torch.float32][k] - grads[torch.float64][k]).norm() / n64[k]) for k in n64}
    er = np.array(sorted(e_ref.values()))
    era_d = {k: float((grads[torch.float32][k] - g_m32[k]).norm() / n64[k]) for k in n64}
    print("reference fp32, largest arithmetic-only errors:", sorted(((v, k) for k, v in era_d.items()), reverse=True)[:5], flush=True)
    era = np.array(sorted(era_d.values()))
    erf = np.array(sorted(float((g_m32[k] - grads[torch.float64][k]).norm() / n64[k]) for k in n64))
    print(f"reference fp32: e median {np.median(er):.2e} p90 {er[int(0.9 * len(er))]:.2e} max {er[-1]:.2e} | e_arith median {np.median(era):.2e} max {era[-1]:.2e} | "
          f"e_flip median {np.median(erf):.2e} max {erf[-1]:.2e} | flips {sum(ref_flips.values())} of {sum(elements.values())}", flush=True)
    del g_m32
    result = {"geometry": {"A": A, "h": h, "w": w, "scale": s, "B": B}, "truth": "fp64 autograd over oracle/lfsr_torch_port.py:distgssr_forward_graph",
              "yardstick": "the same graph in fp32 on stock torch CPU ops (the reference's train.py:256-264 on the CPU)",
              "loss_fp64": losses[torch.float64], "loss_fp32": losses[torch.float32],
              "reference_fp32": {"median": float(np.median(er)), "p90": float(er[int(0.9 * len(er))]), "max": float(er[-1]),
                                 "tensors_beyond_1e-4": int((er > 1e-4).sum()), "mask_flips_vs_fp64": ref_flips, "mask_flips_total": sum(ref_flips.values()),
                                 "decomposition": {"e_arith max": float(era[-1]), "e_arith median": float(np.median(era)), "e_flip max": float(erf[-1]), "e_flip median": float(np.median(erf)),
                                                   "e_flip tensors_beyond_1e-4": int((erf > 1e-4).sum())}},
              "mask_elements": elements, "mask_elements_total": sum(elements.values()), "modes": []}
    M = load_plugin()
    from argparse import Namespace
    xa, la = torch.from_numpy(x).cuda(), torch.from_numpy(label).cuda()
    sel_keys = ("LFSR_CONV3X3", "LFSR_DGRAD3", "LFSR_WGRAD3")
    for name, env, decompose in MODES:
        for k in sel_keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=s))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net = net.to("cuda:0").train()
        out = net(xa, None)
        loss = M.get_loss(None)(out, la, None)
        rt = net._rt
        hf = {k: 0 for k in TH.MASK_KINDS}
        hip_masks = {}
        for i, pre in enumerate(blocks):     # read BEFORE backward (its scratch lives in the same workspace; the saved activations themselves are not overwritten)
            for k in TH.MASK_KINDS:
                m = TH.hip_saved_mask(rt, xa, k, i)
                hf[k] += int((m != masks[torch.float64][pre + k]).sum())
                if decompose:
                    hip_masks[pre + k] = TH.hip_mask_to_ref(m, k, B, A, h, w)
        loss.backward()
        torch.cuda.synchronize()
        g_hip = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}
        rows = [(float((g_hip[k] - grads[torch.float64][k]).norm() / n64[k]), e_ref[k], k) for k in g_hip]
        eh = np.array(sorted(r[0] for r in rows))
        beyond = [(k, e, r) for e, r, k in rows if e > max(1e-4, 3.0 * r)]
        ratio = np.array([e / max(r, 1e-30) for e, r, _ in rows])
        entry = {"mode": name, "env": env, "loss": float(loss.item()),
                 "hip": {"median": float(np.median(eh)), "p90": float(eh[int(0.9 * len(eh))]), "max": float(eh[-1]), "tensors_beyond_1e-4": int((eh > 1e-4).sum())},
                 "tensors_breaking_max(1e-4,3*e_ref)": len(beyond), "worst_ratio_e_hip_over_e_ref": float(ratio.max()), "median_ratio": float(np.median(ratio)),
                 "mask_flips_vs_fp64": hf, "mask_flips_total": sum(hf.values()),
                 "worst": [{"param": k, "e_hip": e, "e_ref": r} for e, r, k in sorted(rows, reverse=True)[:8]],
                 "breaking": [{"param": k, "e_hip": e, "e_ref": r} for k, e, r in sorted(beyond, key=lambda t: -t[1])[:12]],
                 "per_tensor": {k: [e, r] for e, r, k in rows}}
        print(f"{name}\n   HIP median {entry['hip']['median']:.2e} p90 {entry['hip']['p90']:.2e} max {entry['hip']['max']:.2e} | >1e-4: {entry['hip']['tensors_beyond_1e-4']}"
              f" | breaking max(1e-4, 3 e_ref): {len(beyond)} | e_hip/e_ref median {entry['median_ratio']:.1f} max {entry['worst_ratio_e_hip_over_e_ref']:.1f}"
              f" | flips {sum(hf.values())} (ref fp32: {sum(ref_flips.values())})", flush=True)
        if decompose:
            g_m, _, _ = cpu_graph(torch.float64, force=hip_masks)
            e_arith = {k: float((g_hip[k] - g_m[k]).norm() / n64[k]) for k in g_hip}
            e_flip = {k: float((g_m[k] - grads[torch.float64][k]).norm() / n64[k]) for k in g_hip}
            ea, ef = np.array(sorted(e_arith.values())), np.array(sorted(e_flip.values()))
            entry["decomposition"] = {
                "e_arith (HIP vs fp64 graph with the HIP forward's LeakyReLU decisions)": {"median": float(np.median(ea)), "p90": float(ea[int(0.9 * len(ea))]), "max": float(ea[-1]),
                                                                                          "argmax": max(e_arith, key=e_arith.get), "tensors_beyond_1e-4": int((ea > 1e-4).sum())},
                "e_flip (fp64 graph with HIP decisions vs fp64 graph)": {"median": float(np.median(ef)), "p90": float(ef[int(0.9 * len(ef))]), "max": float(ef[-1]),
                                                                        "argmax": max(e_flip, key=e_flip.get), "tensors_beyond_1e-4": int((ef > 1e-4).sum())},
                "per_tensor [e_arith, e_flip]": {k: [e_arith[k], e_flip[k]] for k in g_hip}}
            print(f"   decomposition: e_arith median {np.median(ea):.2e} p90 {ea[int(0.9 * len(ea))]:.2e} max {ea[-1]:.2e} | e_flip median {np.median(ef):.2e} p90 {ef[int(0.9 * len(ef))]:.2e} max {ef[-1]:.2e}", flush=True)
            del g_m, hip_masks
        result["modes"].append(entry)
        del net
    for k in sel_keys:
        os.environ.pop(k, None)
    print(f"reference fp32: median {result['reference_fp32']['median']:.2e} p90 {result['reference_fp32']['p90']:.2e} max {result['reference_fp32']['max']:.2e}")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(result, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
