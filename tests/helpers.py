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
