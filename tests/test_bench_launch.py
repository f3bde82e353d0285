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


def test_shard_tables_and_launcher_lines_for_2_4_8_gpus():
    """What the driver will run at N = 2, 4, 8: the plan's shard table is shard_range's (contiguous, ordered, sizes
    within one of each other, 100,000 genomes in total) and the launcher line is the documented one."""
    sys.path.insert(0, ROOT)
    from miekki_amd.shard import shard_range
    for n in (2, 4, 8):
        p = plan("--gpus", str(n), "--steps", "3", "--warmup", "1")
        assert p["scaling"] == "strong" and len(p["shards"]) == n
        assert p["shards"] == [list(shard_range(100_000, r, n)) for r in range(n)]
        sizes = [b - a for a, b in p["shards"]]
        assert sum(sizes) == 100_000 and max(sizes) - min(sizes) <= 1 and p["shards"][0][0] == 0
        L = p["launcher"]
        assert L[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in L and f"--nproc-per-node={n}" in L
        assert L[L.index("--master-addr") + 1] == "127.0.0.1" and L[L.index("--master-port") + 1].isdigit()
        tail = L[L.index(BENCH) + 1:] if BENCH in L else L[-6:]
        assert tail == ["--gpus", str(n), "--steps", "3", "--warmup", "1"]            # the ranks get the same arguments
    p = plan("--gpus", "8", "--genomes", "100003")                                     # a remainder goes to the first ranks
    assert [b - a for a, b in p["shards"]] == [12501] * 3 + [12500] * 5


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
