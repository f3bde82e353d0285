"""a fingerprint matrix larger than its HBM budget.  The reference's answer to a
collection that outgrows fast memory was to keep columns zlib-compressed in RAM (compress_index /
decompress_index, Miekki.cpp:863-877, inflated per batch by get_minimizers, 881-898); here the
partition rows beyond the budget live in page-locked host memory and whole cold partition ranges
are streamed through staging buffers in HBM.  MIEKKI_HBM_MATRIX_MIB forces a tiny budget.

Every result of the cold context is compared with the CPU ORACLE on the same seeded genomes (index
stream, raw score rows, hits) -- not with another HIP context: the staged path presents a shifted
base pointer as "the matrix", splits the range the hot / cold boundary falls into, and re-lays hot
and cold rows out when the matrix grows, none of which an all-in-HBM context exercises."""
import hashlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope="module")
def hip():
    import miekki_amd
    return miekki_amd


def masked_sha(raw):
    raw = np.frombuffer(bytes(raw), np.uint8).copy()
    raw[32] = 0                                                  # uninitialised jaccard_estimation byte
    return hashlib.sha256(raw.tobytes()).hexdigest()


def same_hits(got, o, rows, nres, ms, mi):
    for q, row in enumerate(rows):
        want = o.filter_results(row, nres, ms, mi)
        assert [(x.genome, x.matches) for x in got[q]] == [(w[0], w[1]) for w in want], q
        np.testing.assert_allclose([x.jaccard for x in got[q]], [w[2] for w in want], rtol=RTOL, atol=0)
        np.testing.assert_allclose([x.intersection for x in got[q]], [w[3] for w in want], rtol=RTOL, atol=0)


def check_against_oracle(ix, o, seqs, rng, nq_stream=700):
    """index stream, plain / dense / long score rows, streamed and by-count hits of `ix` vs the oracle `o`"""
    G = len(seqs)
    assert ix.index_size == o.index_size == G
    np.testing.assert_array_equal(ix.sketch_size, o.sketch_size)
    np.testing.assert_array_equal(ix.genome_size, o.genome_size)
    assert masked_sha(b"".join(ix.serialize())) == masked_sha(o.serialize().tobytes())     # export of hot and cold rows, Bloom, sizes
    qs = []
    for q in range(nq_stream):                                   # >= 512 short queries: ranges by partition, cold ranges streamed
        g = int(rng.integers(0, G)); s = seqs[g]
        off = int(rng.integers(0, max(1, len(s) - 1300)))
        qs.append(s[off:off + 300 + q % 900])
    rows = o.query_sequences(qs)
    got, act = ix.query(qs, 10, 3, 5.0)
    same_hits(got, o, rows, 10, 3, 5.0)
    assert [int(a) for a in act[:40]] == [o.query_sequence(s)[1] for s in qs[:40]]         # active_minimizer, Miekki.cpp:318-340
    got16, _ = ix.query(qs[:16], 10, 3, 5.0)                     # a handful of queries: the small-set schedule over cold rows
    same_hits(got16, o, rows[:16], 10, 3, 5.0)
    np.testing.assert_array_equal(ix.query_sequences(qs[:48]), rows[:48])                   # plain kernel, dense rows out
    long_q = [seqs[7][:9000], seqs[8], seqs[G - 1], seqs[G // 2][100:6000]]                 # sparse long path and dense (whole-genome) path
    lrows = o.query_sequences(long_q)
    np.testing.assert_array_equal(ix.query_sequences(long_q), lrows)
    same_hits(ix.query(long_q, 5, 3, 5.0)[0], o, lrows, 5, 3, 5.0)
    return qs, rows


# (bits per fingerprint, genomes, what the 1 MiB budget does at k = 21, h = 12, four partition ranges of 1024 rows)
SHAPES = [
    (8, 600, "pitch 1 KiB: rows [0, 1024) hot = range 0 exactly"),
    (8, 1100, "pitch 2 KiB from 1,025 genomes on: the boundary moves from row 1024 to row 512, inside range 0"),
    (16, 600, "pitch 1 KiB, then 2 KiB: boundary at row 512, inside range 0, after it was at 1024"),
]


@pytest.mark.parametrize("fpb,G,what", SHAPES)
def test_matrix_beyond_its_hbm_budget_against_oracle(hip, monkeypatch, tmp_path, fpb, G, what):
    from oracle import oracle as orc
    k, h = 21, 12
    seqs = [synth.genome_bases(9000 + g, 0, 12_000 + 37 * (g % 600)) for g in range(G)]
    seqs[5] = seqs[5][:40].lower() + seqs[5][40:6000] + b"NNNNNNNNNN" + seqs[5][6000:]      # a genome that is not plain ACGT
    monkeypatch.setenv("MIEKKI_SLAB_MIB", "1")                   # four partition ranges at h = 12: the slab schedule is in play
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "1")             # 1 MiB of a 4 ... 8 MiB matrix stays in HBM
    o = orc.OracleMiekki(k, h, fpb, 32, 10)
    cold = hip.Miekki(k, h, fpb, 32, 10)
    loaded = None
    rng = np.random.default_rng(5 + G + fpb)
    try:
        step = 150
        for i in range(0, G, step):                              # no reserve: the matrix is re-laid out (hot and cold rows) as it grows
            cold.insert_sequences(seqs[i:i + step])
            o.insert_sequences(seqs[i:i + step])
            if i == step:                                        # a streamed query BETWEEN growth steps: the staging buffers and their
                mid = [s[200:900] for s in seqs[:2 * step:3]] * 6                # events must survive the re-layout that follows
                assert len(mid) >= 512
                same_hits(cold.query(mid, 10, 3, 5.0)[0], o, o.query_sequences(mid), 10, 3, 5.0)
        qs, rows = check_against_oracle(cold, o, seqs, rng)
        # dump -> load under the same budget: import of hot and cold rows
        cold.dump_disk(str(tmp_path / "cold.gz"))
        loaded = hip.Miekki.load(str(tmp_path / "cold.gz"))
        same_hits(loaded.query(qs, 10, 3, 5.0)[0], o, rows, 10, 3, 5.0)
        assert masked_sha(b"".join(loaded.serialize())) == masked_sha(o.serialize().tobytes())
        # append past the capacity AFTER streamed queries, then query again (the staging events)
        more = [synth.genome_bases(20_000 + g, 0, 9000 + 11 * g) for g in range(500)]
        cold.insert_sequences(more)
        o.insert_sequences(more)
        check_against_oracle(cold, o, seqs + more, rng, nq_stream=520)
    finally:
        cold.close()
        if loaded is not None:
            loaded.close()


def test_reserved_cold_matrix_against_oracle(hip, monkeypatch):
    """mk_reserve up front (what the CLI does): one layout, boundary inside a range, 2-byte fingerprints,
    and a budget that leaves most of the matrix cold."""
    from oracle import oracle as orc
    k, h, G, fpb = 15, 13, 700, 16
    seqs = [synth.genome_bases(31_000 + g, 0, 7000 + 13 * g) for g in range(G)]
    monkeypatch.setenv("MIEKKI_SLAB_MIB", "1")                   # eight ranges of 1024 rows
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "3")             # pitch 2 KiB: 1,536 hot rows = one and a half ranges of 8,192 rows
    o = orc.OracleMiekki(k, h, fpb, 32, 10)
    ix = hip.Miekki(k, h, fpb, 32, 10)
    try:
        ix.reserve(G)
        for i in range(0, G, 64):
            ix.insert_sequences(seqs[i:i + 64])
        o.insert_sequences(seqs)
        check_against_oracle(ix, o, seqs, np.random.default_rng(77))
    finally:
        ix.close()


@pytest.mark.parametrize("fpb,rate_ppm,G", [(8, 1000, 1500), (16, 10_000, 700), (8, 0, 1100)])
def test_packed_cold_rows_against_oracle(hip, monkeypatch, tmp_path, fpb, rate_ppm, G):
    """The cold rows PACKED (mk_index_compress: per 1,024 genomes of a row a "differs from the genome before" bit each +
    the differing fingerprints; cold.hip) -- a collection of related strains, species by species in the list, so that
    there is something to pack (rate 0: every strain equals its species -- runs of 25 equal fingerprints; the 2-byte case
    at 1 %: most fingerprints differ and some rows are kept as they are).  Every query path must answer from the packed
    rows what the oracle answers (slab ranges staged packed and expanded in HBM, small sets, the plain and dense kernels
    over packed windows); exports, a dump, an append and a second compress go through the automatic unpacking."""
    from oracle import oracle as orc
    k, h, strains = 21, 12, 25
    L_ = 14_000
    seqs = [synth.strain_device(g, strains, rate_ppm, 0, L_) for g in range(G)]
    monkeypatch.setenv("MIEKKI_SLAB_MIB", "1")                   # four partition ranges at h = 12
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "1")             # 1 MiB of a 4 ... 8 MiB matrix stays in HBM
    o = orc.OracleMiekki(k, h, fpb, 32, 10)
    o.insert_sequences(seqs)
    ix = hip.Miekki(k, h, fpb, 32, 10)
    loaded = None
    rng = np.random.default_rng(11 + G)
    try:
        ix.reserve(G)
        ix.insert_synthetic_strains(0, G, L_, strains, rate_ppm)  # the device's generator of the same strains
        raw, packed = ix.compress_index()
        assert raw > 0 and packed < raw, (raw, packed)
        if rate_ppm <= 1000:
            assert packed < 0.5 * raw                                # 0.1 % divergence: most fingerprints repeat the genome before
        assert ix.compress_index() == (raw, packed)                  # packed already
        qs, rows = check_against_oracle(ix, o, seqs, rng)            # (its export of the index stream unpacks ...)
        assert ix.compress_index()[1] == packed                      # ... and packing again gives the same bytes
        got, _ = ix.query(qs, 10, 3, 5.0)                            # streamed from packed rows this time
        same_hits(got, o, rows, 10, 3, 5.0)
        got16, _ = ix.query(qs[:16], 10, 3, 5.0)
        same_hits(got16, o, rows[:16], 10, 3, 5.0)
        np.testing.assert_array_equal(ix.query_sequences(qs[:48]), rows[:48])          # plain kernel over packed windows
        long_q = [seqs[7][:9000], seqs[8], seqs[G - 1]]
        np.testing.assert_array_equal(ix.query_sequences(long_q), o.query_sequences(long_q))   # dense kernel over packed windows
        ix.dump_disk(str(tmp_path / "packed.gz"))                    # unpacks by itself (Miekki.cpp:662-664 does the same)
        loaded = hip.Miekki.load(str(tmp_path / "packed.gz"))
        assert masked_sha(b"".join(loaded.serialize())) == masked_sha(o.serialize().tobytes())
        assert ix.compress_index()[1] == packed
        more = [synth.strain_device(G + g, strains, rate_ppm, 0, L_) for g in range(70)]
        ix.insert_sequences(more)                                    # an append into a packed index unpacks it first
        o.insert_sequences(more)
        raw2, packed2 = ix.compress_index()
        assert packed2 < raw2
        qs2 = [s[100:1200] for s in (seqs + more)[::3]][:520] + [more[3][:800]] * 8
        same_hits(ix.query(qs2, 10, 3, 5.0)[0], o, o.query_sequences(qs2), 10, 3, 5.0)
        ix.decompress_index()
        assert masked_sha(b"".join(ix.serialize())) == masked_sha(o.serialize().tobytes())
    finally:
        ix.close()
        if loaded is not None:
            loaded.close()


def test_independent_genomes_are_left_unpacked(hip, monkeypatch):
    """Unrelated genomes: no row shrinks (profiles/r3_column_entropy.txt), mk_index_compress measures that and leaves the
    rows as they are."""
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "1")
    ix = hip.Miekki(21, 12, 8, 32, 10)
    try:
        ix.reserve(1200)
        ix.insert_synthetic(0, 1200, 9000)
        raw, packed = ix.compress_index()
        assert raw > 0 and packed == raw
        hits, _ = ix.query([synth.genome_bases(5, 100, 1500)], 3, 3, 1.0)
        assert hits[0] and hits[0][0].genome == 5
    finally:
        ix.close()
    monkeypatch.delenv("MIEKKI_HBM_MATRIX_MIB")                      # (the budget is read when the context is made)
    ix = hip.Miekki(21, 12, 8, 32, 10)                               # and an index without cold rows: nothing to do
    try:
        ix.insert_synthetic(0, 100, 9000)
        assert ix.compress_index() == (0, 0)
    finally:
        ix.close()


def test_persistent_set_across_reserve_and_compress(hip, monkeypatch):
    """a query set prepared for the small-set schedule (pieces cut by count) while every row is in HBM, then
    mk_reserve sends rows to host memory (no new index generation: the set is not prepared again), mk_index_compress packs
    them, and the same set runs again: the scan must unpack BEFORE it takes the matrix's addresses -- it read rows beyond
    the hot ones out of the hot allocation.  Checked against the oracle's filter over the oracle's score rows."""
    import ctypes as C
    import torch
    from oracle import oracle as orc
    from miekki_amd import lib as L, distributed as mkd
    k, h, G, strains, rate, L_ = 21, 10, 240, 20, 1000, 9000
    monkeypatch.setenv("MIEKKI_HBM_MATRIX_MIB", "1")             # 1,024 rows x 1 KiB fit; at 2 KiB a row (after the reserve) 512 do
    seqs = [synth.strain_device(g, strains, rate, 0, L_) for g in range(G)]
    o = orc.OracleMiekki(k, h, 8, 32, 10)
    o.insert_sequences(seqs)
    ix = hip.Miekki(k, h, 8, 32, 10)
    lib = L.load_library()
    qset = C.c_void_p()
    try:
        ix.insert_synthetic_strains(0, G, L_, strains, rate)
        assert ix.compress_index() == (0, 0)                       # nothing cold yet
        qs = [seqs[(7 * q) % G][50 * q:50 * q + 700 + 20 * q] for q in range(24)]
        rows = o.query_sequences(qs)
        ptrs, lens = L.seq_arrays(qs)
        L.check(lib.mk_qset_upload(ix._h, ptrs, lens, len(qs), C.byref(qset)))
        cap, nres = 256, 10                                        # (strains of one species tie: many entrants; cap >= G never overflows)
        d_count = torch.zeros(len(qs), dtype=torch.int32, device="cuda")
        d_cand = torch.zeros(len(qs) * cap * 24, dtype=torch.uint8, device="cuda")

        def run():
            L.check(lib.mk_qset_run(ix._h, qset, nres, 3, 5.0, cap, d_count.data_ptr(), d_cand.data_ptr()))
            L.check(lib.mk_sync(ix._h))
            hits, over = mkd.merge_candidates(d_count.cpu().numpy()[None], d_cand.cpu().numpy()[None], cap, nres)
            assert not over.any()
            for q, row in enumerate(rows):
                want = o.filter_results(row, nres, 3, 5.0)
                assert [(int(x["genome"]), int(x["matches"])) for x in hits[q]] == [(w[0], w[1]) for w in want], q
        run()
        before = ix.stats()["scan_slab_launches"]
        ix.reserve(1100)                                           # pitch 2 KiB: rows [512, 1024) leave for host memory
        raw, packed = ix.compress_index()
        assert raw > 0 and packed < raw, (raw, packed)
        run()                                                      # the same prepared set over packed cold rows
        assert ix.stats()["scan_slab_launches"] > before
        assert ix.compress_index()[1] == packed                    # (the in-place read unpacked them; packing again is the same bytes)
        run()
    finally:
        if qset:
            lib.mk_qset_free(ix._h, qset)
        ix.close()
