"""GPU: `python bench.py --gpus 2` end to end for the three workloads that have a multi-rank leg, rehearsed on ONE device: both ranks on
cuda:0 (LFSR_BENCH_ONE_DEVICE=1) with gloo as the process-group backend (RCCL refuses two ranks on one device).  What runs is the product path of
every rank -- fresh child processes spawned by bench.py's own launcher, barrier + max-over-ranks clock, the flat-bucket all-reduce of the training
step, the crop -> gather -> place exchange of the sharded scene -- only the transport differs from the 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["infer", "train", "lft"])
def test_two_ranks_one_device(workload):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LFSR_BENCH_ONE_DEVICE="1", LFSR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-workloads",
           "--workload", workload, "--rank-timeout", "300"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "patches/s"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    per_step = {"infer": 64, "train": 16, "lft": 64}[workload]        # patches all ranks process per step
    assert abs(line["value"] - per_step / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["scaling"] == ("strong" if workload == "lft" else "weak")
    if workload == "train":
        assert 0.0 < line["loss"] < 1.0
