"""a fingerprint matrix larger than its HBM budget.  The reference's answer to a
collection that outgrows fast memory was to keep columns zlib-compressed in RAM (compress_index /
decompress_index, Miekki.cpp:863-877); here the partition rows beyond the budget live in page-locked
host memory and the slab schedule streams whole cold partition ranges through a staging buffer.
MIEKKI_HBM_MATRIX_MIB forces a tiny budget: every result must equal that of an all-in-HBM context."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import miekki_amd
    return miekki_amd


@pytest.mark.parametrize("fpb", [8, 16])
def test_matrix_beyond_its_hbm_budget(hip, monkeypatch, tmp_path, fpb):
    k, h, G = 21, 12, 600
    seqs = [synth.genome_bases(9000 + g, 0, 12_000 + 37 * g) for g in range(G)]
    monkeypatch.setenv("MIEKKI_SLAB_MIB", "1")                   # four partition ranges at h = 12: the slab schedule is in play
    full = hip.Miekki(k, h, fpb, 32, 10)
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "1")             # 1 MiB of a 4 MiB (8 MiB at 2 bytes) matrix stays in HBM
    cold = hip.Miekki(k, h, fpb, 32, 10)
    loaded = None
    try:
        full.insert_sequences(seqs)
        for i in range(0, G, 150):                               # no reserve: the matrix is re-laid out (hot and cold rows) as it grows
            cold.insert_sequences(seqs[i:i + 150])
        assert b"".join(cold.serialize()) == b"".join(full.serialize())      # export of hot and cold rows, Bloom, sizes
        rng = np.random.default_rng(5)
        qs = []
        for q in range(700):                                     # >= 512 short queries: ranges by partition, cold ranges streamed
            g = int(rng.integers(0, G)); o = int(rng.integers(0, 10_000))
            qs.append(seqs[g][o:o + 300 + q % 900])
        want, wact = full.query(qs, 10, 3, 5.0)
        got, gact = cold.query(qs, 10, 3, 5.0)
        assert got == want and (gact == wact).all()
        got16, _ = cold.query(qs[:16], 10, 3, 5.0)               # a handful: ranges by count, cold rows read in place
        assert got16 == want[:16]
        np.testing.assert_array_equal(cold.query_sequences(qs[:40]), full.query_sequences(qs[:40]))     # plain kernel
        long_q = [seqs[7][:9000], seqs[8]]                       # sparse long path and dense (whole-genome) path
        np.testing.assert_array_equal(cold.query_sequences(long_q), full.query_sequences(long_q))
        assert cold.query(long_q, 5, 3, 5.0)[0] == full.query(long_q, 5, 3, 5.0)[0]
        # dump -> load under the same budget: import of hot and cold rows
        cold.dump_disk(str(tmp_path / "cold.gz"))
        loaded = hip.Miekki.load(str(tmp_path / "cold.gz"))
        assert loaded.query(qs, 10, 3, 5.0)[0] == want
        assert b"".join(loaded.serialize()) == b"".join(full.serialize())
    finally:
        full.close(); cold.close()
        if loaded is not None:
            loaded.close()
