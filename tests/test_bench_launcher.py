"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` without a launcher environment; SURVEY 8e, the reference has nothing to match:
train.py:166 keeps a vestigial local_rank).  The children here run bench.py's launcher test hook -- no torch, no GPU: what is tested is the parent:
it hands every rank its environment, relays rank 0's line, and when a rank fails or hangs it stops the others and exits non-zero within its timeout."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(hook, *extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LFSR_BENCH_TEST_RANK"] = hook
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", *extra], env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.monotonic() - t0


def test_ranks_get_their_environment_and_rank0_line_is_relayed():
    p, _ = run("ok")
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["ok"] is True and line["world"] == 3 and int(line["port"]) > 0


def test_failed_rank_stops_the_others():
    p, el = run("fail:1")              # ranks 0 and 2 would sleep for 10 minutes
    assert p.returncode != 0
    assert "rank 1 exited with status 3" in p.stderr
    assert el < 60


def test_hung_rank_is_killed_at_the_timeout():
    p, el = run("hang:2", "--rank-timeout", "3")
    assert p.returncode != 0
    assert "still running after" in p.stderr
    assert el < 60
