"""GPU: a runtime's forward captured into a HIP graph (capi.GraphedForward) replays the eager launches: bit-equal outputs, also for a second input of the same shape."""
import json
import os

import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    key = {"distgssr": "DistgSSR", "epit": "EPIT", "lft": "LFT"}[name]
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"][key]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.DistgSSRRuntime(5, 4) if name == "distgssr" else capi.ModelRuntime(name, 5, 4, 5 if name == "epit" else 4, 64)
    rt.load_state([(k, torch.from_numpy(v).cuda()) for k, v in sd.items()], torch.device("cuda"))
    return rt


@pytest.mark.parametrize("name,B", [("epit", 1), ("distgssr", 2), ("lft", 1)])
def test_graph_replay_equals_eager(name, B):
    rt = _runtime(name)
    gf = capi.GraphedForward(rt)
    for seed in (3, 4):                                   # the second input replays the graph captured for the first
        x = torch.from_numpy(synth_input((B, 1, 160, 160), seed=seed)).cuda()
        y_graph = gf(x).clone()
        y_eager = rt.forward(x)
        torch.cuda.synchronize()
        assert torch.equal(y_graph, y_eager), (name, seed)
    assert len(gf.graphs) == 1
