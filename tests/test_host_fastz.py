"""host/fastz.cpp (CRC-32 by carry-less multiplication, whole-buffer inflate, the Huffman-only deflate of the index
columns) against zlib, on the CPU: tests/helpers/fastz_check.cpp feeds zlib's streams of every level / strategy / flush
form to the reader here, the writer's streams to zlib's reader, gzip members with every optional header field, truncated
and bit-flipped inputs -- once optimised, once under AddressSanitizer + UBSan (nothing may be read or written out of bounds
whatever the input)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "helpers", "fastz_check.cpp"), os.path.join(ROOT, "host", "fastz.cpp")]


def build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "host"), *flags, "-o", exe, *SRC, "-lz"], check=True)
    return exe


def test_fastz_against_zlib(tmp_path):
    r = subprocess.run([build(tmp_path, "fastz_check", ["-O2"])], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    assert int(r.stdout.split()[1]) > 50_000


def test_fastz_under_sanitizers(tmp_path):
    exe = build(tmp_path, "fastz_check_san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr[-2000:]
