"""bench.py on the GPU box: the one-GPU line at a small size and the 2-rank rehearsal of the
default (strong-scaling) multi-GPU workload -- gloo collectives, both ranks on GPU 0, started
from a cold shell the way the driver starts it (plain `python bench.py --gpus 2 ...`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_one_gpu_line_and_two_rank_rehearsal_agree():
    common = ["--genomes", "192", "--queries", "2000", "--h", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = run_bench("--gpus", "1", *common)
    two = run_bench("--gpus", "2", "--rehearse", *common)
    for r, n in ((one, 1), (two, 2)):
        assert r["n_gpus"] == n and r["scaling"] == "strong" and r["config"]["genomes_total"] == 192
        assert r["unit"] == "comparisons/s" and r["value"] > 0
        assert r["roofline"]["frac"] == pytest.approx(r["roofline"]["achieved"] / r["roofline"]["peak"])
        assert r["roofline"]["hbm_floor_bytes"] == (1 << 20) * r["config"]["genomes_per_gpu"]
        assert r["merge"]["overflowed_queries"] == 0
        # every query comes out with its source genome on top, and the device heap is the host heap
        assert r["check"]["top_hit_is_source_genome_of_all_queries"] == r["check"]["queries"] == 2000
        assert r["check"]["device_heap_equals_host_heap_of_strided_sample"] == r["check"]["strided_sample"] == 2000
    assert two["config"]["genomes_per_gpu"] == 96
    # same problem, same gate: identical active partitions, hence identical comparisons per step
    assert one["config"]["active_partitions_per_query"] == two["config"]["active_partitions_per_query"]
    assert two["merge"]["gather_bytes_per_rank"] == 2000 * (two["merge"]["cap"] + 1) * 8
    assert one["merge"]["gather_bytes_per_rank"] == 0
