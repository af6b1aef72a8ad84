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


@pytest.mark.parametrize("geom", [(5, 3, 3, 128, 64, 160, 132), (3, 2, 5, 64, 32, 64, 150), (2, 1, 1, 8, 4, 3, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_crop_place_equal_integrate(geom, dtype):
    """lfsr_lf_crop_tiles / lfsr_lf_place_tiles (what the sharded dispatcher exchanges) against the oracle and against lfsr_lf_integrate, bit for bit,
    for any split of the patch list, incl. the ragged crop at (h, w) and an empty shard"""
    A, numU, numV, pz, stride, h, w = geom
    n = numU * numV
    sub = torch.from_numpy(np.random.default_rng(5).standard_normal((n, A * pz, A * pz)).astype(np.float32)).to(dtype).cuda()
    want = capi.lf_integrate(sub.reshape(numU, numV, A * pz, A * pz), A, pz, stride, h, w)
    tiles = capi.lf_crop_tiles(sub, A, pz, stride)
    assert np.array_equal(tiles.cpu().numpy(), O.lf_crop_tiles(sub.cpu().numpy(), A, pz, stride))
    for cut in (0, 1, n // 2, n):
        out = torch.full((A, A, h, w), float("nan"), dtype=dtype, device="cuda")
        capi.lf_place_tiles(capi.lf_crop_tiles(sub[:cut], A, pz, stride), out, A, numU, numV, 0, stride)
        capi.lf_place_tiles(capi.lf_crop_tiles(sub[cut:], A, pz, stride), out, A, numU, numV, cut, stride)
        assert torch.equal(out, want)
    with pytest.raises(capi.LfsrError):
        capi.lf_place_tiles(tiles, torch.empty((A, A, h, w), dtype=dtype, device="cuda"), A, numU, numV, 1, stride)   # first + count beyond the list
