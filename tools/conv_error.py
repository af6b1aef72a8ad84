#!/usr/bin/env python3
"""Round-off of the HIP DistgSSR forward against the fp64 oracle's golden output, for the Winograd (default) and the direct
(LFSR_CONV3X3=halo) form of the 3x3 convs.  usage: python tools/conv_error.py"""
import os, sys
os.environ.setdefault("LFSR_LAB", "1")   # (this tool drives the library's A/B selectors, live only under LFSR_LAB)
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import model_case, psnr
from lfsr_amd import capi

for tag in ("a5h8s4", "a3h6w8s2"):
    case, sd, x, gold = model_case("DistgSSR", tag)
    A, s = case["A"], case["s"]
    for env in ("", "halo"):
        if env: os.environ["LFSR_CONV3X3"] = env
        else: os.environ.pop("LFSR_CONV3X3", None)
        rt = capi.DistgSSRRuntime(A, s)
        rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
        y = rt.forward(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        ref = gold[tag + "_out"].astype(np.float64)
        d = np.abs(y - ref)
        print(f"{tag} conv={'winograd' if not env else 'direct  '}: max|err| {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  PSNR(hip, ref) {psnr(y, ref):.1f} dB  (|ref| max {np.abs(ref).max():.2f})")
