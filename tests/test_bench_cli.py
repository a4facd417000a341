"""bench.py's rank handling, on the CPU: --gpus N starts N ranks (or insists that the launcher did), the ranks meet over
gloo, shard the channels without overlap, and rank 0 prints one JSON line.  (--dry-run: control flow only, no GPU work.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def last_json(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout + p.stderr
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_that_meet_and_shard():
    p = run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert d["n_gpus"] == 2 and d["config"]["rendezvous"] == "gloo"
    assert d["channels_total"] == 2 * 65536                       # weak scaling: per-GPU work fixed
    assert abs(d["max_wall"] - 2e-3) < 1e-9                       # MAX over ranks, not rank 0's own


def test_million_workload_is_strong_scaling_over_the_ranks():
    p = run(["--gpus", "3", "--dry-run", "--workload", "million"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert d["n_gpus"] == 3 and d["channels_total"] == 1 << 20    # blocks differ by at most one channel, none lost
    assert d["config"]["channels_per_gpu"] == (1 << 20) // 3 + 1


def test_gpus_must_match_the_launcher():
    p = run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    p = run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0"}, drop=())
    assert p.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in p.stderr


def test_without_a_gpu_every_rank_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this box has a GPU")
    p = run(["--gpus", "2", "--no-cpu-baseline"])
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "rank exit codes" in p.stderr
