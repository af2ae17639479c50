"""CPU: `python bench.py --gpus N` from a plain shell (no torchrun environment) spawns its own ranks, they rendezvous on
127.0.0.1 and see world_size N (the bench used to run one rank and print n_gpus 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                     # rank 0 only
    assert json.loads(lines[0])["n_gpus"] == n


@pytest.mark.timeout(60)
def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-launch"], capture_output=True, text=True, env=env, timeout=50)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.timeout(90)
def test_bench_fails_fast_when_one_rank_dies_at_startup():
    """one rank exiting non-zero must take the job down at once (the others are blocked in the rendezvous), not leave it
    to the driver's time limit.  Rank 1 exits with code 3 before init_process_group (EGOMI_BENCH_FAIL_RANK); the parent must return
    non-zero within seconds and leave no child behind."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(EGOMI_BENCH_FAIL_RANK="1", EGOMI_BENCH_RDV_TIMEOUT="600")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--backend", "gloo", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=80)
    dt = time.time() - t0
    assert r.returncode != 0, (r.stdout, r.stderr[-500:])
    assert dt < 30, dt
    assert "rank 1 exited with code 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]                 # no result line from a broken job
