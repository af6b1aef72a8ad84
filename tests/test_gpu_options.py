"""GPU: the library's two option mechanisms (csrc/options.cpp).
 * lfsr_set_arithmetic(LFSR_ARITH_F32) selects exactly the fp32-MFMA kernels the A/B selectors LFSR_EPI=wino / LFSR_ROWGEMM=f32 select (same bits), the default differs from them
   (three-term bf16 kernels) and both stay within the operator tolerance of the fp64 oracle of the reference layers (DistgSSR.py:91-100).
 * the LFSR_* selectors are live only in a process started with LFSR_LAB set: in a process without it they change nothing (checked in a child process)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lfsr_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _epi_and_fuse(seed=3):
    B, A, h, w = 1, 5, 16, 16
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * A * A * h * w, 64, generator=g).cuda()
    w1 = capi.pack_conv_weight((torch.randn(32, 64, 1, 25, generator=g) * 0.03).cuda())
    w2 = capi.pack_conv_weight((torch.randn(160, 32, 1, 1, generator=g) * 0.15).cuda())
    wf = capi.pack_conv_weight((torch.randn(64, 144, 1, 1, generator=g) * 0.1).cuda())
    cat = torch.randn(B * A * A * h * w, 144, generator=g).cuda()
    out = torch.zeros((x.shape[0], 64), device="cuda")
    capi.epiconv_hv(x, w1, w2, B, A, h, w, 0.1, out, 0, 32)
    y = capi.pointwise(cat, 144, wf, 64, slope=0.1)
    return out.clone(), y.clone()


def test_arithmetic_api_selects_the_fp32_kernels(monkeypatch):
    for k in ("LFSR_EPI", "LFSR_ROWGEMM"):
        monkeypatch.delenv(k, raising=False)
    assert capi.load().lfsr_get_arithmetic() == capi.ARITH_DEFAULT
    e_def, f_def = _epi_and_fuse()
    capi.set_arithmetic(capi.ARITH_F32)
    try:
        e_api, f_api = _epi_and_fuse()
    finally:
        capi.set_arithmetic(capi.ARITH_DEFAULT)
    monkeypatch.setenv("LFSR_EPI", "wino"); monkeypatch.setenv("LFSR_ROWGEMM", "f32")
    e_env, f_env = _epi_and_fuse()
    monkeypatch.delenv("LFSR_EPI"); monkeypatch.delenv("LFSR_ROWGEMM")
    assert torch.equal(e_api, e_env) and torch.equal(f_api, f_env)            # the API and the lab selectors pick the same kernels
    assert not torch.equal(e_def, e_api) and not torch.equal(f_def, f_api)    # ... which are not the default ones
    assert float((e_def - e_api).abs().max()) < 1e-4 and float((f_def - f_api).abs().max()) < 1e-4
    with pytest.raises(capi.LfsrError):
        capi.set_arithmetic(7)


def test_selectors_are_dead_without_lfsr_lab():
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from tests.test_gpu_options import _epi_and_fuse\n"
        "import os\n"
        "a = _epi_and_fuse()\n"
        "os.environ['LFSR_EPI'] = 'wino'; os.environ['LFSR_ROWGEMM'] = 'f32'\n"
        "b = _epi_and_fuse()\n"
        "print('SAME' if torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) else 'DIFFERENT')\n" % ROOT)
    for lab, want in ((None, "SAME"), ("1", "DIFFERENT")):
        env = {k: v for k, v in os.environ.items() if k not in ("LFSR_LAB", "LFSR_EPI", "LFSR_ROWGEMM")}
        if lab:
            env["LFSR_LAB"] = lab
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-1500:]
        assert p.stdout.strip().splitlines()[-1] == want, (lab, p.stdout, p.stderr[-500:])
