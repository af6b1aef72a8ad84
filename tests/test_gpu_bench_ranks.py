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
        assert line["config"]["gemm_arithmetic"]["epiconv"] == "bf16x3" and line["config"]["gemm_arithmetic"]["conv3x3"] == "f32"
    # what the ranks themselves saw of the job: the collective's own count of participants, the backend, every rank's clock and device
    mg = line["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["backend"].startswith("gloo")
    assert len(mg["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in mg["per_rank_ms_per_step"])
    assert max(mg["per_rank_ms_per_step"]) <= line["ms_per_step"] * (1 + 1e-6) + 1e-3           # the line's clock is the max over ranks
    assert len({d["pid"] for d in mg["per_rank_device"]}) == 2 and all(d["cuda_index"] == 0 for d in mg["per_rank_device"])
    if workload == "lft":
        assert line["config"]["scene"] == {"size": [5, 5, 128, 128], "patches": 64}


def test_lft_scene_size_flag():
    """--scene-size: the 5x5x256x256 scene is 256 patches (the strong-scaling leg of an 8-GPU node takes 512: 1024 patches, 128 per GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--workload", "lft", "--scene-size", "256"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["scene"] == {"size": [5, 5, 256, 256], "patches": 256}
    assert abs(line["value"] - 256 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
