"""Drop-in check in the other direction: the REAL reference binary loads an index
file written by OUR writer -- host/gzpar.cpp, the parallel gzip layer of the
`miekki` binary, exercised here through the stand-alone `mkgz` filter (no GPU
needed) -- and answers queries from it exactly as from its own dump.

The reference part needs oracle/_ref/Miekki (built from /root/reference by
`make -C oracle ref`), so it runs in the authoring container and skips elsewhere.
The stream itself comes from the oracle, which test_oracle_golden.py pins
byte-for-byte to the reference's dump."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "Miekki")
MKGZ = os.path.join(ROOT, "miekki_amd", "mkgz")


def mkgz(args, data=None, env=None):
    r = subprocess.run([MKGZ, *args], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    return r.stdout


@pytest.mark.parametrize("size", [0, 1, 70_000, (32 << 20) - 1, 32 << 20, (96 << 20) + 12345])
def test_parallel_gzip_round_trips_and_is_plain_gzip(tmp_path, size):
    rng = np.random.default_rng(size % 1000)
    data = (rng.integers(0, 4, size, dtype=np.uint8) + 200).tobytes()          # fingerprint-like bytes
    path = str(tmp_path / "x.gz")
    mkgz(["c", path, "6"], data)
    assert mkgz(["d", path, "5"]) == data                                      # our parallel reader
    assert gzip.decompress(open(path, "rb").read()) == data                    # any gzip reader
    members = open(path, "rb").read().count(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x04\x03\x0c\x00MK\x08\x00")
    assert members == max(1, -(-size // (32 << 20)))
    # and files that are NOT ours go through the generic path: one-member gzip, plain bytes
    other = str(tmp_path / "y.gz")
    open(other, "wb").write(gzip.compress(data, 1))
    assert mkgz(["d", other, "4"]) == data
    open(other, "wb").write(data)
    if size >= 2 and data[:2] != b"\x1f\x8b":
        assert mkgz(["d", other, "4"]) == data


def test_reader_reads_members_in_parallel(tmp_path):
    """The reader inflates members out of buffers it preads, several at a time: same bytes."""
    rng = np.random.default_rng(11)
    data = (rng.integers(0, 50, (70 << 20) + 4321, dtype=np.uint8) + 100).tobytes()
    path = str(tmp_path / "x.gz")
    mkgz(["c", path, "6"], data)
    assert gzip.decompress(open(path, "rb").read()) == data
    assert mkgz(["d", path, "5"]) == data


@pytest.mark.parametrize("stored", [False, True])
def test_index_writer_forms_are_plain_gzip_too(tmp_path, stored):
    """The dump's special forms (host/index_io.cpp through gzpar): the 39-byte head as a member of its own, blocks filled
    in place, Huffman-only deflate for the fingerprint columns -- or stored blocks, MIEKKI_DUMP_LEVEL=0 -- and the
    ready-made member for every whole block of zeros (the unreachable part of the reference's 1 GiB Bloom filter): our
    reader, zlib and therefore the reference's zstr read the same stream back."""
    rng = np.random.default_rng(3)
    cols = (rng.integers(0, 60, (70 << 20) + 999, dtype=np.uint8) + 190).tobytes()      # fingerprint-like columns
    data = bytes(39) + cols + bytes(8 * 1000) + b"\x01\x02" * 4000 + bytes((100 << 20) + 17) + b"tail" * 1000
    path = str(tmp_path / "i.gz")
    env = dict(os.environ, MKGZ_STORED="1") if stored else dict(os.environ)
    r = subprocess.run([MKGZ, "i", path, "6"], input=data, stdout=subprocess.PIPE, env=env)
    assert r.returncode == 0
    assert mkgz(["d", path, "5"]) == data
    raw = open(path, "rb").read()
    assert gzip.decompress(raw) == data
    if stored:
        assert len(raw) > len(cols)                                             # the columns went in as they are
    else:
        assert len(raw) < 0.85 * len(cols)                                      # uniform over 60 values: 5.9 of 8 bits per byte
    # the 100 MiB of zeros (less what the block before them took): ready-made members (a few KB each, identical), not deflate runs
    sizes, indexed, at = [], 0, 0
    while at < len(raw):                                                        # member by member: 12 + XLEN + stream + 8
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 16] == b"MK\x08\x00"
        xlen, payload = int.from_bytes(raw[at + 10:at + 12], "little"), int.from_bytes(raw[at + 16:at + 24], "little")
        indexed += xlen > 12 and raw[at + 24:at + 26] == b"MH"
        sizes.append(12 + xlen + payload + 8)
        at += sizes[-1]
    assert at == len(raw)
    assert (indexed == 0) if stored else (indexed >= 2)                         # Huffman-only members carry their block index
    assert sum(1 for a, b in zip(sizes, sizes[1:]) if a == b and a < 200_000) >= 1   # identical small members in a row


def test_corrupt_member_is_detected(tmp_path):
    data = bytes(range(256)) * 4000
    path = str(tmp_path / "x.gz")
    mkgz(["c", path, "2"], data)
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(path, "wb").write(bytes(raw))
    r = subprocess.run([MKGZ, "d", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0


@pytest.mark.skipif(not os.path.exists(REF), reason="reference build (oracle/_ref) not present")
def test_reference_loads_an_index_written_by_our_writer(tmp_path, golden_dir):
    case = synth.CASES["messy"]()
    o = orc.OracleMiekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
    o.insert_sequences(case.genome_sequences())
    mkgz(["c", str(tmp_path / "idx.gz"), "8"], o.serialize().tobytes())        # 17 members of 32 MiB
    (tmp_path / "queries.fa").write_bytes(b"".join(h + b"\n" + s + b"\n" for h, s in case.queries))
    r = subprocess.run([REF, "-i", "idx.gz", "-a", "queries.fa", "-o", "out.txt", "-t", "1"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert (tmp_path / "out.txt").read_bytes() == open(os.path.join(golden_dir, "messy_out.txt"), "rb").read()
