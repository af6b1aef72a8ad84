"""Deterministic synthetic weights and inputs (numpy only, no torch RNG).

There are no pretrained checkpoints in the reference tree (``.gitignore:8`` ignores ``*.pth``), so
parity is pinned on synthetic weights.  To keep the committed fixtures small the weights are not
stored: every tensor is regenerated from ``(seed, state_dict key, shape)`` with numpy's PCG64
stream, which is stable across numpy versions.  The golden generator (``tests/golden/make_golden.py``)
loads these tensors into the *reference* model; tests, ``bench.py`` and ``smoke()`` load the very
same tensors into the HIP path and the oracle.

Scale follows PyTorch's default conv/linear init (``kaiming_uniform_(a=sqrt(5))`` == U(-1/sqrt(fan_in),
1/sqrt(fan_in))) so activations stay in the range the reference sees with random-init weights.
"""
import zlib

import numpy as np


def _rng(seed, key):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode("utf-8"))])


def synth_tensor(key, shape, seed=0):
    """One fp32 tensor for state_dict entry ``key`` of shape ``shape``."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(seed, key)
    if len(shape) == 1:
        u = rng.uniform(-1.0, 1.0, size=shape)
        if key.endswith("weight"):          # LayerNorm gain (the only 1-D ``weight`` on the path)
            return (1.0 + 0.1 * u).astype(np.float32)
        return (0.1 * u).astype(np.float32)  # conv / LayerNorm bias
    fan_in = int(np.prod(shape[1:]))
    bound = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def synth_state_dict(spec, seed=0):
    """``spec`` = iterable of ``(key, shape)`` -> ``{key: np.ndarray(float32)}`` in the same order."""
    return {k: synth_tensor(k, s, seed) for k, s in spec}


def synth_input(shape, seed=1):
    """LR luminance patch batch in [0,1) (``utils_datasets.py:49`` feeds [0,1] singles)."""
    return np.random.default_rng(int(seed)).random(size=tuple(shape), dtype=np.float32)
