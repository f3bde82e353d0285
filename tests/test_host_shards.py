"""Host logic of the multi-GPU `miekki` binary that needs no GPU: which devices it would use
(MIEKKI_DEVICES / MIEKKI_DEVICE / every visible one) and how a list is cut into genome shards --
the same rule as miekki_amd.shard.shard_range, which bench.py and the multi-process driver use."""
import os
import subprocess

import pytest

from miekki_amd.shard import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sh") / "shard_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-I", os.path.join(ROOT, "include"), "-o", out,
                    os.path.join(ROOT, "tests", "helpers", "shard_check.cpp"), os.path.join(ROOT, "host", "multi_gpu.cpp"),
                    "-L", os.path.join(ROOT, "miekki_amd"), "-lmiekki_hip", "-lpthread",
                    "-Wl,-rpath," + os.path.join(ROOT, "miekki_amd")], check=True)
    return out


def run(exe, n, parts, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("MIEKKI_DEVICES", "MIEKKI_DEVICE")}
    e.update(env)
    out = subprocess.run([exe, str(n), str(parts)], stdout=subprocess.PIPE, env=e, check=True, timeout=120).stdout.decode().splitlines()
    devs = [int(x) for x in out[0].split()[1:]]
    shards = [tuple(int(x) for x in l.split()[2:]) for l in out[1:] if l.startswith("shard")]
    run.caps = tuple(int(x) for x in [l for l in out if l.startswith("cap")][0].split()[1:])
    return devs, shards


def test_device_list_from_the_environment(exe):
    assert run(exe, 0, 1, MIEKKI_DEVICES="0,0,0")[0] == [0, 0, 0]          # repeated ordinals: several shards on one GPU
    assert run(exe, 0, 1, MIEKKI_DEVICES="3,1")[0] == [3, 1]
    assert run(exe, 0, 1, MIEKKI_DEVICE="2")[0] == [2]
    assert run(exe, 0, 1, MIEKKI_DEVICES="1", MIEKKI_DEVICE="2")[0] == [1]   # the list wins
    assert len(run(exe, 0, 1)[0]) >= 1                                       # every visible GPU (at least the ordinal mk_create will refuse)


def test_shard_tables_are_the_python_rule(exe):
    for n, parts in ((100_000, 8), (9, 2), (7, 8), (0, 4), (12, 5)):
        _, shards = run(exe, n, parts)
        assert shards == [shard_range(n, r, parts) for r in range(parts)]
        assert shards[0][0] == 0 and shards[-1][1] == n


def test_entrant_row_width_is_the_python_rule(exe):
    """host/multi_gpu.hpp: entrant_cap (the `miekki` binary) and miekki_amd/shard.py: entrant_cap (bench.py) are one rule:
    every rank of a run must size its exchange rows alike."""
    from miekki_amd.shard import entrant_cap
    for n in (0, 7, 1000, 12_500, 50_000, 100_000, 4_000_000):
        run(exe, n, 1)
        assert run.caps == (entrant_cap(10, n), entrant_cap(5, n)), n
    assert entrant_cap(10, 12_500) == entrant_cap(10, 50_000) == entrant_cap(10, 100_000) == 128
