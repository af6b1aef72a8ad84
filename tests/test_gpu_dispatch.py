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
