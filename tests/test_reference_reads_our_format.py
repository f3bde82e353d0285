"""Drop-in check in the other direction: the REAL reference binary loads an index
file laid out the way our writer lays it out -- the row-P stream cut into 32 MiB
blocks, each its own gzip member (host/index_io.cpp ParallelGzipWriter) -- and
answers queries from it exactly as from its own dump.

Needs oracle/_ref/Miekki (built from /root/reference by `make -C oracle ref`), so it
runs in the authoring container and skips elsewhere.  The stream itself comes from
the oracle, which test_oracle_golden.py pins byte-for-byte to the reference's dump."""
import gzip
import os
import subprocess

import pytest

import synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "Miekki")


@pytest.mark.skipif(not os.path.exists(REF), reason="reference build (oracle/_ref) not present")
def test_reference_loads_multi_member_gzip_index(tmp_path, golden_dir):
    case = synth.CASES["messy"]()
    o = orc.OracleMiekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
    o.insert_sequences(case.genome_sequences())
    raw = o.serialize().tobytes()
    block = 32 << 20
    with open(tmp_path / "idx.gz", "wb") as f:
        for i in range(0, len(raw), block):
            f.write(gzip.compress(raw[i:i + block], 1))            # one member per block
    (tmp_path / "queries.fa").write_bytes(b"".join(h + b"\n" + s + b"\n" for h, s in case.queries))
    r = subprocess.run([REF, "-i", "idx.gz", "-a", "queries.fa", "-o", "out.txt", "-t", "1"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert (tmp_path / "out.txt").read_bytes() == open(os.path.join(golden_dir, "messy_out.txt"), "rb").read()
