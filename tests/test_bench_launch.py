"""bench.py's launch path without a GPU: argument parsing, the shard table of the default
(strong-scaling) workload, and the self-launch of `--gpus N` -- which must start the ranks as
children and fail cleanly, not hang or crash, when no GPU is there."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def plan(*argv):
    out = subprocess.run([sys.executable, BENCH, "--plan", *argv], stdout=subprocess.PIPE, check=True, timeout=120)
    return json.loads(out.stdout)


def test_default_multi_gpu_workload_is_config3_strong_scaling():
    p = plan("--gpus", "8")
    assert p["scaling"] == "strong"
    assert p["shards"][0] == [0, 12500] and p["shards"][-1] == [87500, 100000]       # 100,000 genomes TOTAL
    assert all(a[1] == b[0] for a, b in zip(p["shards"], p["shards"][1:]))
    assert "torch.distributed.run" in p["launcher"] and "--nproc-per-node=8" in p["launcher"]
    assert p["launcher"][p["launcher"].index("--master-addr") + 1] == "127.0.0.1"
    assert plan("--gpus", "1")["shards"] == [[0, 100000]] and plan("--gpus", "1")["launcher"] is None


def test_weak_scaling_is_opt_in():
    p = plan("--gpus", "4", "--weak", "--genomes-per-gpu", "100000")
    assert p["scaling"] == "weak" and p["shards"][-1] == [300000, 400000]
    p = plan("--gpus", "2", "--genomes-per-gpu", "50")                                # implies --weak
    assert p["scaling"] == "weak" and p["shards"] == [[0, 50], [50, 100]]
    p = plan("--gpus", "3", "--genomes", "10")
    assert p["shards"] == [[0, 4], [4, 7], [7, 10]]


def test_plain_multi_gpu_invocation_self_launches_and_fails_cleanly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by the -m gpu rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--genomes", "8", "--queries", "4",
                        "--steps", "1", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0
    assert b"bench.py needs a GPU" in r.stderr                  # the ranks started and said why they stop
    assert not r.stdout.strip()                                  # and no JSON line was invented
