"""CPU: the plugin mirrors export the reference's surface (get_model / get_loss / weights_init) and the exact state_dict
contract (key names, order, shapes) recorded from the reference models (tests/golden/models.json); forward refuses CPU tensors."""
import importlib
from argparse import Namespace

import pytest
import torch

from lfsr_amd import capi
from tests.helpers import models_meta

PARAMS = {"DistgSSR": 3581568, "EPIT": 1470080, "LFT": 1163392, "LF_InterNet": 5040320}   # README.md:144,174-177 of the reference


@pytest.mark.parametrize("name", sorted(PARAMS))
def test_state_dict_contract(name):
    M = importlib.import_module("lfsr_amd.model.SR." + name)
    assert callable(M.get_model) and callable(M.get_loss) and callable(M.weights_init)
    meta = models_meta()["models"][name]
    for tag, case in list(meta["cases"].items()) + [("full", meta["full"])]:
        net = M.get_model(Namespace(angRes_in=case["A"], angRes_out=case["A"], scale_factor=case["s"]))
        net.apply(M.weights_init)
        got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert got == case["spec"], (name, tag)
    assert sum(p.numel() for p in net.parameters()) == PARAMS[name]
    with pytest.raises(capi.LfsrError):
        with torch.no_grad():
            net(torch.zeros(1, 1, case["A"] * 4, case["A"] * 4), None)      # CPU tensor: no fallback


def test_losses():
    for name in ("DistgSSR", "LFT", "LF_InterNet"):
        M = importlib.import_module("lfsr_amd.model.SR." + name)
        loss = M.get_loss(None)(torch.ones(2, 1, 4, 4), torch.zeros(2, 1, 4, 4), [5, 5])
        assert float(loss) == 1.0
    E = importlib.import_module("lfsr_amd.model.SR.EPIT")
    with pytest.raises((TypeError, IndexError)):                              # upstream quirk kept: out['SR'] on a tensor
        E.get_loss(None)(torch.ones(1, 1, 2, 2), torch.zeros(1, 1, 2, 2))


def test_h5_layout_roundtrip():
    """N4: the transposed on-disk layout of the test scenes (Generate_Data_for_Test.py:88-92 / utils_datasets.py:111-128)"""
    import numpy as np
    from lfsr_amd.utils.h5_layout import from_h5_arrays, to_h5_arrays
    rng = np.random.default_rng(3)
    lr, hr, cc = rng.random((10, 15)), rng.random((20, 30)), rng.random((20, 30, 2))
    s_lr, s_hr, s_cc = to_h5_arrays(lr, hr, cc)
    assert s_lr.shape == (15, 10) and s_hr.shape == (30, 20) and s_cc.shape == (2, 30, 20) and s_cc.dtype == np.float32
    t_lr, t_hr, t_cc = from_h5_arrays(s_lr, s_hr, s_cc)
    assert t_lr.shape == (1, 10, 15) and t_hr.shape == (1, 20, 30) and t_cc.shape == (2, 20, 30)
    assert np.allclose(t_lr[0].numpy(), lr.astype(np.float32)) and np.allclose(t_cc.numpy(), cc.astype(np.float32).transpose(2, 0, 1))
    # degenerate chroma as the reference handles it
    _, _, z = from_h5_arrays(s_lr, s_hr, np.zeros((), dtype=np.float32))
    assert z.shape == (2, 20, 30) and float(z.abs().max()) == 0.0
    _, _, one = from_h5_arrays(s_lr, s_hr, rng.random((20, 30)).astype(np.float32))
    assert one.shape == (1, 20, 30)
