"""CPU: the plugin mirrors export the reference's surface (get_model / get_loss / weights_init) and the exact state_dict
contract (key names, order, shapes) recorded from the reference models (tests/golden/models.json); forward refuses CPU tensors."""
import importlib
import os
import sys
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


def _import_option(argv):
    """option.py parses sys.argv at import time (as the reference's does, option.py:36): import a fresh copy under a patched argv."""
    import importlib.util
    from lfsr_amd import capi
    spec = importlib.util.spec_from_file_location("lfsr_option_under_test", os.path.join(capi._HERE, "option.py"))
    mod = importlib.util.module_from_spec(spec)
    old = sys.argv
    sys.argv = ["train.py"] + argv
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = old
    return mod.args


def test_option_defaults_and_derived_fields():
    """the flag schema of the reference's option.py:4-46: defaults, the derived SR fields and the deleted ``angRes``"""
    a = _import_option([])
    assert (a.task, a.scale_factor, a.model_name, a.batch_size, a.lr, a.decay_rate, a.n_steps, a.gamma, a.epoch) == ("SR", 2, "LFT", 4, 2e-4, 0, 15, 0.5, 51)
    assert (a.device, a.num_workers, a.local_rank, a.use_pre_ckpt, a.use_masked_pretrain, a.mask_ratio) == ("cuda:0", 2, 0, True, True, 0.3)
    assert (a.path_pre_pth, a.data_name, a.path_for_train, a.path_for_test, a.path_log) == ("./pth/", "ALL", "./data_for_training/", "./data_for_test/", "./log/")
    assert (a.angRes_in, a.angRes_out, a.patch_size_for_test, a.stride_for_test, a.minibatch_for_test) == (5, 5, 32, 16, 1)
    assert not hasattr(a, "angRes")
    b = _import_option(["--angRes", "7", "--scale_factor", "4", "--model_name", "DistgSSR", "--use_pre_ckpt", "False", "--mask_ratio", "0.1",
                        "--batch_size", "8", "--device", "cuda:3", "--local_rank", "2"])
    assert (b.angRes_in, b.angRes_out, b.scale_factor, b.model_name, b.batch_size, b.device, b.local_rank, b.mask_ratio) == (7, 7, 4, "DistgSSR", 8, "cuda:3", 2, 0.1)
    assert b.use_pre_ckpt is True        # type=bool quirk kept: any non-empty string is truthy, as upstream
    c = _import_option(["--task", "RE"])
    assert not hasattr(c, "angRes_in") and not hasattr(c, "angRes")   # derived fields exist for task SR only (option.py:39-44)


def test_harness_helper_aliases_are_the_capi_kernels():
    """utils/utils.py exposes the reference's names (LFdivide / LFintegrate / ImageExtend, train.py:5 ``from utils.utils import *``)"""
    from lfsr_amd import capi
    from lfsr_amd.utils import utils as U
    for name in ("LFdivide", "LFintegrate", "ImageExtend"):
        assert callable(getattr(U, name))
    with pytest.raises(capi.LfsrError):
        U.LFdivide(torch.zeros(160, 160), 5, 32, 16)      # CPU tensor: the HIP path has no CPU fallback
