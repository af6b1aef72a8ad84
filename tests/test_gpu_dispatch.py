"""GPU: full-scene path (LFdivide -> batched DistgSSR forward -> LFintegrate) through the sharded dispatcher
equals the reference's per-patch loop (train.py:300-318) evaluated patch by patch on the same HIP model, and one
patch is checked against the oracle."""
import numpy as np
import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.dispatch import sr_scene
from lfsr_amd.synth import synth_input
from oracle import lfsr_oracle as O
from tests.helpers import model_case

pytestmark = pytest.mark.gpu


def test_scene_matches_per_patch_loop():
    case, sd, _, _ = model_case("DistgSSR", "full")
    rt = capi.DistgSSRRuntime(5, 4)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
    A, h0, w0 = 5, 40, 33                                        # -> 3 x 3 = 9 patches, ragged crop
    lr = torch.from_numpy(synth_input((A * h0, A * w0), seed=3)).cuda()
    net = lambda x, info=None: rt.forward(x.contiguous())
    out = sr_scene(net, lr, A, 4, minibatch=4)
    assert tuple(out.shape) == (A, A, h0 * 4, w0 * 4)
    # the reference's loop: minibatch 1, stitched by the oracle on the host
    sub = capi.lf_divide(lr, A, 32, 16)
    n1, n2 = sub.shape[:2]
    outs = [rt.forward(sub[i, j][None, None].contiguous()).cpu().numpy()[0, 0] for i in range(n1) for j in range(n2)]
    ref = O.lf_integrate(np.stack(outs).reshape(n1, n2, 640, 640), A, 128, 64, h0 * 4, w0 * 4)
    assert np.array_equal(out.cpu().numpy(), ref)               # batching and stitching change no bit
    # and one patch against the oracle itself
    p0 = sub[1, 1][None, None].cpu().numpy()
    o0 = O.distgssr_forward(p0, sd, 5, 4)
    assert np.abs(outs[1 * n2 + 1] - o0[0, 0]).max() < 1e-4


def _per_patch_reference(rt, lr, A, s):
    sub = capi.lf_divide(lr, A, 32, 16)
    n1, n2 = sub.shape[:2]
    outs = torch.stack([rt.forward(sub[i, j][None, None].contiguous())[0, 0] for i in range(n1) for j in range(n2)])
    return capi.lf_integrate(outs.reshape(n1, n2, A * 32 * s, A * 32 * s).contiguous(), A, 32 * s, 16 * s, lr.shape[0] // A * s, lr.shape[1] // A * s), n1 * n2


@pytest.mark.parametrize("model", ["DistgSSR", "LFT"])
@pytest.mark.parametrize("h0,w0,npatch", [(128, 128, 64), (125, 125, 64), (108, 156, 70)])
def test_scene_baseline_sizes(model, h0, w0, npatch):
    """configs[4]'s scene sizes (5x5x128^2, 5x5x125^2, 5x5x108x156 -> 64 / 64 / 70 patches) through sr_scene with
    minibatch 32 equal the reference's patch-by-patch loop (train.py:300-318) bit for bit, DistgSSR and LFT."""
    A, s = 5, 4
    case, sd, _, _ = model_case(model, "full")
    if model == "DistgSSR":
        rt = capi.DistgSSRRuntime(A, s)
    else:
        rt = capi.ModelRuntime("lft", A, s, 4, 64)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda", 0))
    lr = torch.from_numpy(synth_input((A * h0, A * w0), seed=3)).cuda()
    out = sr_scene(lambda x, info=None: rt.forward(x.contiguous()), lr, A, s, minibatch=32)
    ref, n = _per_patch_reference(rt, lr, A, s)
    assert n == npatch and tuple(out.shape) == (A, A, h0 * s, w0 * s)
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref)
