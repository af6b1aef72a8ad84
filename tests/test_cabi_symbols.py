"""CPU: the C-ABI library loads and exports every symbol include/lfsr_hip.h declares (no compute calls)."""
import os
import re

import pytest

from lfsr_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "lfsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lfsr_[a-z0-9_]+)\s*\(", src)))


def test_library_built():
    assert os.path.exists(capi.LIB_PATH), "build with __graft_entry__.build()"


def test_exports_every_declared_symbol():
    lib = capi.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in lfsr_hip.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(capi.SIGNATURES.keys()) == syms


def test_version_string():
    assert b"gfx950" in capi.load().lfsr_version()


def test_host_only_entry_points():
    """Entry points that touch no device memory can be exercised without a GPU."""
    import ctypes as C
    lib = capi.load()
    nu, nv = C.c_int(0), C.c_int(0)
    for (h0, w0, eu, ev) in [(128, 128, 8, 8), (125, 125, 8, 8), (108, 156, 7, 10), (40, 33, 3, 3)]:
        assert lib.lfsr_lf_divide(None, None, 5, h0, w0, 32, 16, 4, C.byref(nu), C.byref(nv), None) == 0
        assert (nu.value, nv.value) == (eu, ev)      # SURVEY 8c expected counts
    assert lib.lfsr_lf_divide(None, None, 5, 0, 8, 32, 16, 4, None, None, None) == -1
    assert lib.lfsr_packed_weight_floats(64, 64, 9) == (9 + 16 + 36) * 64 * 64   # direct pack + the F(2x2,3x3) and F(4x4,3x3) Winograd-domain copies
    assert lib.lfsr_packed_weight_floats(64, 128, 9) == 9 * 64 * 128
    assert lib.lfsr_packed_weight_floats(400, 16, 1) == 416 * 16
    ctx = C.c_void_p()
    assert lib.lfsr_distgssr_create(C.byref(ctx), 5, 4, 4, 4, 64) == 0
    assert lib.lfsr_distgssr_packed_bytes(ctx) > 3581568 * 4
    assert lib.lfsr_distgssr_workspace_bytes(ctx, 32, 32, 32) > 0
    assert lib.lfsr_distgssr_load_param(ctx, b"nope", None, 0, None) == -1
    lib.lfsr_distgssr_destroy(ctx)
    assert lib.lfsr_distgssr_create(C.byref(ctx), 5, 4, 4, 4, 32) == -1   # channels fixed at 64


def test_no_cpu_fallback():
    import torch
    with pytest.raises(capi.LfsrError):
        capi.sai2macpi(torch.zeros(1, 1, 10, 10), 5)
