"""Host ingest (host/fasta_reader.cpp): ordered parallel FASTA reading with pooled buffers.

The sequence of a file is what Miekki.cpp:559-567 builds with getline: every line not
starting with '>' appended to one string.  Checked against a Python statement of that on
plain, gzip, multi-member gzip, CRLF, unterminated, empty and missing files, for several
thread counts and read-ahead windows (order must be list order for every -t).
"""
import gzip
import re
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def reference_sequence(text: bytes) -> bytes:
    return b"".join(l for l in text.split(b"\n") if not l.startswith(b">"))


@pytest.fixture(scope="module")
def dumper(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rd") / "reader_dump")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "helpers", "reader_dump.cpp"),
                    os.path.join(ROOT, "host", "fasta_reader.cpp"), os.path.join(ROOT, "host", "fastz.cpp"), "-lz", "-lpthread"], check=True)
    return exe


def test_reader_matches_getline_semantics(dumper, tmp_path):
    rng = np.random.default_rng(5)

    def bases(n):
        return bytes(rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), n))

    texts = []
    for i in range(40):
        n = int(rng.integers(0, 30000))
        body = bases(n)
        w = int(rng.choice([60, 80, 1000000]))
        lines = [b">seq%d some description" % i] + [body[j:j + w] for j in range(0, len(body), w)]
        if i % 5 == 0:
            lines.insert(len(lines) // 2, b">contig2")                  # multi-FASTA: concatenated
        t = (b"\r\n" if i % 7 == 3 else b"\n").join(lines)
        if i % 3:
            t += b"\n"                                                  # else: unterminated last line
        texts.append(t)
    texts += [b"", b"\n\n", b">only a header\n", b"ACGT", b">h\n\n\nAC\n\nGT\n"]
    # line ends and headers at and around the packer's 32-character steps (and what a 64-byte-window form would meet), a '>' inside a line (a character, not a header), headers longer than a window, headers in a row,
    # a header right behind an empty line, no newline at the end of a header
    for n in (62, 63, 64, 65, 127, 128, 129):
        texts.append(b">h\n" + bases(n) + b"\n" + bases(5) + b"\n")
        texts.append(bases(n) + b"\n>" + b"x" * 70 + b"\n" + bases(n + 1))
        texts.append(bases(n - 3) + b"A>C\n>second\n>third\n\n>fourth\n" + bases(9) + b"\n\n" + bases(n))
    texts += [bases(40) + b"\n>unterminated header", b">" + b"y" * 200 + b"\n" + bases(64) + b"\n" + bases(64) + b"\n",
              b"\n" * 70 + bases(3) + b"\n" * 64 + bases(2), bases(64) + b"\n>" , b">\n" + bases(1)]
    names, want = [], []
    for i, t in enumerate(texts):
        fn = str(tmp_path / f"f{i}.fa")
        if i % 4 == 1:
            fn += ".gz"
            with gzip.open(fn, "wb") as f:
                f.write(t)
        elif i % 4 == 2:                                                # two gzip members back to back
            fn += ".gz"
            cut = len(t) // 2
            with open(fn, "wb") as f:
                f.write(gzip.compress(t[:cut]) + gzip.compress(t[cut:]))
        else:
            with open(fn, "wb") as f:
                f.write(t)
        names.append(fn)
        want.append(reference_sequence(t))
        if i % 9 == 4:
            names.append(str(tmp_path / f"missing{i}.fa"))
            want.append(None)
    lst = str(tmp_path / "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(names) + "\n")
    for threads, window in ((1, 1), (3, 2), (8, 64)):
        out = subprocess.run([dumper, lst, str(threads), str(window)], check=True, stdout=subprocess.PIPE).stdout.decode().split("\n")
        for i, w in enumerate(want):
            ex, ln, h, failed = out[i].split()
            if w is None:
                assert ex == "0", i
            else:
                assert (ex, int(ln), int(h, 16), failed) == ("1", len(w), fnv1a(w), "0"), (i, names[i], threads)
    # packed items (the reader packs while it parses): same sequences with every non-ACGT character as an
    # exception, the first 32 characters kept as they came, "dirty" exactly where there is an exception
    out = subprocess.run([dumper, lst, "5", "16", "0", str(len(want)), "packed"], check=True, stdout=subprocess.PIPE).stdout.decode().split("\n")
    for i, w in enumerate(want):
        f = out[i].split()
        if w is None:
            assert f[0] == "0", i
            continue
        norm = bytes(c if c in b"ACGT" else ord("?") for c in w)
        assert (f[0], int(f[1]), int(f[2], 16), f[3]) == ("1", len(w), fnv1a(norm), "0"), (i, names[i], "packed")
        assert f[4] == ("1" if norm != w else "0")
        assert (f[5] if len(f) > 5 else "") == w[:32].hex()
    # the portable (non-AVX2) packer gives the same
    out2 = subprocess.run([dumper, lst, "2", "4", "0", str(len(want)), "packed"], check=True, stdout=subprocess.PIPE,
                          env=dict(os.environ, MIEKKI_PACK_SCALAR="1")).stdout.decode().split("\n")
    assert out2[:len(want)] == out[:len(want)]
    # an allocator that runs dry after 100 kB (the page-lock limit): later buffers come from malloc, same
    # sequences, and every allocator-owned buffer goes back through the allocator's release hook
    out = subprocess.run([dumper, lst, "4", "8", "100000"], check=True, stdout=subprocess.PIPE).stdout.decode().split("\n")
    for i, w in enumerate(want):
        ex, ln, h, failed = out[i].split()
        if w is not None:
            assert (ex, int(ln), int(h, 16), failed) == ("1", len(w), fnv1a(w), "0"), (i, "dry allocator")
    assert out[len(want)].startswith("done live=0")
    # destroyed after three items while the workers are parked on the read-ahead bound: must return
    out = subprocess.run([dumper, lst, "8", "2", "0", "3"], check=True, stdout=subprocess.PIPE, timeout=60).stdout.decode().split("\n")
    assert out[3].startswith("done live=0")


def test_unreadable_and_truncated_files_are_flagged_not_read_as_empty(dumper, tmp_path):
    good = b">g\nACGTACGTAC\n"
    (tmp_path / "good.fa").write_bytes(good)
    gz = gzip.compress(b">t\n" + b"ACGT" * 5000 + b"\n")
    (tmp_path / "trunc.fa.gz").write_bytes(gz[:len(gz) // 2])               # cut inside the deflate stream
    (tmp_path / "dir.fa").mkdir()                                           # exists, cannot be read as a file
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(str(tmp_path / n) for n in ("good.fa", "trunc.fa.gz", "dir.fa", "good.fa")) + "\n")
    out = subprocess.run([dumper, str(lst), "2", "4"], check=True, stdout=subprocess.PIPE, timeout=60).stdout.decode().split("\n")
    assert out[0].split()[3] == "0" and out[3].split()[3] == "0" and int(out[0].split()[1]) == 10
    assert out[1].split()[0] == "1" and out[1].split()[3] == "1"             # truncated gzip: failed, not empty
    assert out[2].split()[0] == "1" and out[2].split()[3] == "1"             # a directory: failed


def test_gzipd_files_are_shared_between_the_device_and_the_readers_in_whole_units(dumper, tmp_path):
    """host/fasta_reader.hpp raw_unit / raw_units_ahead: a unit of the list goes raw (for mk_gz_unpack) while the device has
    room, to the readers' zlib otherwise; a unit is never split between the two (a plain file in a device unit is parsed
    here and does not use up the device's room), raw items are the files' bytes, the others the packed sequences."""
    rng = np.random.default_rng(11)
    texts, names = [], []
    for i in range(60):
        body = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(100, 5000))))
        t = b">g%d\n" % i + b"\n".join(body[j:j + 70] for j in range(0, len(body), 70)) + b"\n"
        plain = i in (5, 6, 23)
        fn = tmp_path / ("g%d.fa" % i if plain else "g%d.fa.gz" % i)
        fn.write_bytes(t if plain else gzip.compress(t, 6))
        texts.append(t); names.append(str(fn))
    (tmp_path / "l.txt").write_text("\n".join(names) + "\n")
    unit = 4
    # ("sink": the same through a RawSink -- the files of a device unit are put piece by piece into a batch, the raw items hold
    # no bytes: what the driver does with mk_gz_open / mk_gz_stage / mk_gz_put)
    for threads, ahead, how in ((1, 2, "share"), (4, 2, "share"), (8, 1, "share"), (3, 1000, "share"), (1, 2, "sink"), (4, 2, "sink"), (8, 1, "sink"), (5, 1000, "sink")):
        out = subprocess.run([dumper, str(tmp_path / "l.txt"), str(threads), "16", "0", str(len(names)), f"{how}:{unit}:{ahead}"],
                             check=True, stdout=subprocess.PIPE, timeout=120).stdout.decode().splitlines()
        assert out[-1].startswith("done") and (how == "share" or out[-1].endswith("lent=0")), out[-1]
        assert len(out) == len(names) + 1
        kinds = []
        for i, line in enumerate(out[:-1]):
            f = line.split()
            if f[0] == "raw":
                blob = open(names[i], "rb").read()
                assert (int(f[1]), int(f[2], 16)) == (len(blob), fnv1a(blob)), (threads, ahead, i)
                kinds.append("raw")
            else:
                seq = reference_sequence(texts[i])
                assert int(f[1]) == len(seq) and int(f[2], 16) == fnv1a(seq), (threads, ahead, i)
                kinds.append("plain" if names[i].endswith(".fa") else "host")
        for u in range(0, len(names), unit):
            ks = {k for k in kinds[u:u + unit] if k != "plain"}
            assert len(ks) <= 1, (threads, ahead, u, kinds[u:u + unit])       # never split
        assert kinds[0] == "raw"                                               # the device is empty at the start
        if ahead == 1000:
            assert "host" not in kinds                                         # room for everything: all raw
        else:
            # (the helper gives raw items back a unit at a time, so the device's room comes back: raw units keep coming)
            assert "raw" in kinds[len(kinds) // 2:], (threads, ahead, kinds)


def test_runs_of_a_device_unit_go_up_as_one_span(dumper, tmp_path):
    """With plenty of files a reader takes runs of eight: the files of a run that belong to one device unit are read into
    one lent piece, laid out as the batch's input is (zeros between them), and put with one call -- their items appear when
    the span has been put, still in list order; a plain file, a missing one and a file larger than a piece inside a run do
    not disturb their neighbours."""
    rng = np.random.default_rng(23)
    texts, names = [], []
    for i in range(400):
        n = int(rng.integers(100, 3000)) if i != 77 else 60_000                  # (one file larger than a lent piece)
        body = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)) if i != 77 else bytes(rng.integers(65, 90, n, dtype=np.uint8))
        t = b">g%d\n" % i + b"\n".join(body[j:j + 70] for j in range(0, len(body), 70)) + b"\n"
        plain = i in (9, 130, 131, 399)
        fn = tmp_path / ("g%d.fa" % i if plain else "g%d.fa.gz" % i)
        if i != 200:                                                              # (listed, not there)
            fn.write_bytes(t if plain else gzip.compress(t, 6))
        texts.append(t); names.append(str(fn))
    (tmp_path / "l.txt").write_text("\n".join(names) + "\n")
    for threads, unit in ((2, 16), (5, 32), (1, 8)):
        out = subprocess.run([dumper, str(tmp_path / "l.txt"), str(threads), "64", "0", str(len(names)), f"sink:{unit}:1000:span"],
                             check=True, stdout=subprocess.PIPE, timeout=120).stdout.decode().splitlines()
        assert out[-1].startswith("done") and out[-1].endswith("lent=0"), out[-1]
        assert len(out) == len(names) + 1
        raw = 0
        for i, line in enumerate(out[:-1]):
            f = line.split()
            if i == 200:
                assert f[0] == "0"
            elif f[0] == "raw":
                blob = open(names[i], "rb").read()
                assert (int(f[1]), int(f[2], 16)) == (len(blob), fnv1a(blob)), (threads, unit, i)
                raw += 1
            else:
                seq = reference_sequence(texts[i])
                assert int(f[1]) == len(seq) and int(f[2], 16) == fnv1a(seq), (threads, unit, i)
        assert raw == 400 - 5, raw
        spans, span_files = (int(x) for x in re.search(r"spans=(\d+) of (\d+) files", out[-1]).groups())
        assert spans > 0 and span_files >= 2 * spans, out[-1]                     # (runs of several files did go as spans)
