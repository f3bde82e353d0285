"""The bootstrap file of the one-process-per-GPU form (host/rendezvous.cpp): rank 0 leaves the
communicator's id in a file the other ranks read.  It must not be mistaken for a file a crashed run left behind, must not
be planted, and must not be written through a link."""
import os
import stat
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("rv") / "rendezvous_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", out, os.path.join(ROOT, "tests", "helpers", "rendezvous_check.cpp"),
                    os.path.join(ROOT, "host", "rendezvous.cpp"), "-lpthread"], check=True)
    return out


def fetch(exe, path, nonce, n, timeout_ms):
    r = subprocess.run([exe, "fetch", str(path), nonce, str(n), str(timeout_ms)], stdout=subprocess.PIPE, timeout=60)
    return r.returncode, r.stdout.decode().strip()


def publisher(exe, path, nonce, payload, hold_ms):
    p = subprocess.Popen([exe, "publish", str(path), nonce, payload, str(hold_ms)], stdout=subprocess.PIPE)
    assert p.stdout.readline().strip() == b"published"
    return p


def test_a_live_rank_zero_is_read_and_a_dead_one_is_not(exe, tmp_path):
    path = tmp_path / "comm.id"
    # a run that crashed: its rank 0 published and died, the file stayed
    p = publisher(exe, path, "run-A", "OLDOLDOLD", 0)
    p.wait()
    assert path.exists() and stat.S_IMODE(os.stat(path).st_mode) == 0o600
    rc, out = fetch(exe, path, "run-A", 9, 300)
    assert rc == 1 and "left behind" in out, out
    # the new run: ranks > 0 start first and wait; rank 0 replaces the stale file; they read the NEW id
    waiter = subprocess.Popen([exe, "fetch", str(path), "run-A", "9", "20000"], stdout=subprocess.PIPE)
    time.sleep(0.3)
    p = publisher(exe, path, "run-A", "NEWNEWNEW", 3000)
    assert waiter.communicate(timeout=30)[0].decode().strip() == "got NEWNEWNEW"
    p.wait()


def test_another_runs_file_garbage_and_planted_links_are_not_followed(exe, tmp_path):
    path = tmp_path / "comm.id"
    p = publisher(exe, path, "run-B", "BBBBBBBBB", 4000)
    rc, out = fetch(exe, path, "run-C", 9, 300)                       # another run's nonce: not ours
    assert rc == 1 and "nonce" in out, out
    rc, out = fetch(exe, path, "run-B", 9, 2000)
    assert (rc, out) == (0, "got BBBBBBBBB")
    p.kill(); p.wait()
    path.write_bytes(b"MKCOMM2\n" + bytes(300))                       # garbage with the right magic
    rc, out = fetch(exe, path, "run-B", 9, 300)
    assert rc == 1 and "checksum" in out, out
    # a link planted under the name: the publisher replaces the LINK, the file it pointed at is untouched; a reader that
    # finds a link does not follow it
    victim = tmp_path / "victim.txt"
    victim.write_bytes(b"precious")
    path.unlink()
    os.symlink(victim, path)
    rc, out = fetch(exe, path, "run-B", 9, 300)
    assert rc == 1
    p = publisher(exe, path, "run-B", "CCCCCCCCC", 3000)
    assert victim.read_bytes() == b"precious" and not os.path.islink(path)
    assert fetch(exe, path, "run-B", 9, 2000) == (0, "got CCCCCCCCC")
    p.kill(); p.wait()
