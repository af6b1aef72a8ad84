"""GPU: LF_InterNet forward (BASELINE config 0 geometry: 5x5 x2 on a 32x32 patch) through the C ABI vs the numpy oracle and
the reference's golden outputs."""
import sys

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from oracle import lfsr_oracle as O
from tests.helpers import model_case, psnr

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def runtime(case, sd):
    rt = capi.ModelRuntime("internet", case["A"], case["s"], 4, 4)
    rt.load_state([(k, dev(v)) for k, v in sd.items()], torch.device("cuda", 0))
    return rt


@pytest.mark.parametrize("tag", ["a5h8s2", "a3h6w8s4"])
def test_internet_small_vs_golden_and_oracle(tag):
    case, sd, x, npz = model_case("LF_InterNet", tag)
    y = runtime(case, sd).forward(dev(x)).cpu().numpy()
    gold = npz[tag + "_out"]
    ref = O.internet_forward(x, sd, case["A"], case["s"])
    tol = 1e-4 * max(1.0, np.abs(gold).max())
    assert np.abs(y - ref).max() < tol
    assert np.abs(y - gold).max() < tol


def test_internet_full_patch():
    case, sd, x, npz = model_case("LF_InterNet", "full")       # 5x5 views of 32x32, x2 -> (1,1,320,320)
    y = runtime(case, sd).forward(dev(x)).cpu().numpy()
    assert y.shape == (1, 1, 320, 320)
    assert np.abs(y[:, :, ::8, ::8] - npz["full_sample"]).max() < 1e-4 * max(1.0, np.abs(npz["full_sample"]).max())
    ref = O.internet_forward(x, sd, 5, 2)
    assert psnr(y, ref) >= 80.0


def test_internet_plugin_surface():
    import importlib
    from argparse import Namespace
    sys.path.insert(0, capi._HERE)
    try:
        M = importlib.import_module("model.SR.LF_InterNet")
    finally:
        sys.path.remove(capi._HERE)
    case, sd, x, npz = model_case("LF_InterNet", "a3h6w8s4")
    net = M.get_model(Namespace(angRes_in=3, angRes_out=3, scale_factor=4))
    assert [k for k in net.state_dict()] == [k for k, _ in case["spec"]]
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to("cuda:0").eval()
    with torch.no_grad():
        y = net(dev(x), None)
    assert np.abs(y.cpu().numpy() - npz["a3h6w8s4_out"]).max() < 1e-4 * max(1.0, np.abs(npz["a3h6w8s4_out"]).max())
