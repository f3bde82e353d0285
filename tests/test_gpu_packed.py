"""SURVEY 8f row N2, packed ingest: mk_index_append_packed (2 bits per base + exception bits + the
first 32 characters) must build exactly the index mk_index_append builds from the characters --
and both go through the same device pipeline (build.hip), so these cases pin that pipeline to the
ORACLE: the asymmetric strand codes of anything that is not ACGT (utils.cpp:31-49, 107-125), the
seed's own rules (utils.cpp:252-272: lower case accepted, any other character zeroes it), lengths
around k and around the 32-base words, both fingerprint widths."""
import hashlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import miekki_amd
    return miekki_amd


def masked(raw):
    raw = np.frombuffer(bytes(raw), np.uint8).copy()
    raw[32] = 0                                                  # uninitialised jaccard_estimation byte
    return hashlib.sha256(raw.tobytes()).hexdigest()


def messy_genomes(k, rng):
    base = [synth.genome_bases(700 + i, 0, n) for i, n in enumerate(
        [k, k + 1, k + 2, 31, 32, 33, 63, 64, 65, 4095 + k, 4096 + k, 4097 + k, 8192 + k, 20_000, 60_001, 130_000])]
    base = [b for b in base if len(b) >= k]
    out = list(base)
    g = bytearray(base[-3])
    g[5] = ord("N")                                              # N inside the seed: the whole seed becomes zero
    out.append(bytes(g))
    g = bytearray(base[-3])
    g[:20] = bytes(g[:20]).lower()                               # lower case inside the seed: accepted there
    g[200:260] = bytes(g[200:260]).lower()                       # ... and an exception everywhere else
    g[4090:4100] = b"NNNNNNNNNN"                                 # across a workgroup's segment boundary
    g[9000] = ord("-"); g[9001] = 0; g[9002] = 255
    out.append(bytes(g))
    g = bytearray(base[-2])
    g[-40:] = b"N" * 40                                          # exceptions up to the last position
    g[k - 2] = ord("n"); g[k - 1] = ord("n")                     # last seed position invalid, first rolling position an exception
    out.append(bytes(g))
    junk = np.frombuffer(b"ACGTNacgtnRYKM-", np.uint8)
    out.append(bytes(rng.choice(junk, 30_000)))                  # mostly exceptions
    out.append(b"ACGT" * 9000)                                   # repetitive: bins overflow, the batch is redone from characters
    return out


@pytest.mark.parametrize("k,h,fpb", [(31, 14, 8), (21, 12, 16), (9, 8, 8), (15, 20, 8), (27, 17, 16)])
def test_packed_ingest_against_oracle(hip, k, h, fpb):
    from oracle import oracle as orc
    rng = np.random.default_rng(k * 31 + h)
    seqs = messy_genomes(k, rng)
    o = orc.OracleMiekki(k, h, fpb, 32, 10)
    o.insert_sequences(seqs)
    want = masked(o.serialize().tobytes())
    from_chars = hip.Miekki(k, h, fpb, 32, 10)
    from_packed = hip.Miekki(k, h, fpb, 32, 10)
    mixed = hip.Miekki(k, h, fpb, 32, 10)
    try:
        from_chars.insert_sequences(seqs)
        assert masked(b"".join(from_chars.serialize())) == want
        for i in range(0, len(seqs), 7):                         # several batches, pipelined
            from_packed.insert_sequences_packed(seqs[i:i + 7])
        np.testing.assert_array_equal(from_packed.sketch_size, o.sketch_size)
        np.testing.assert_array_equal(from_packed.genome_size, o.genome_size)
        assert masked(b"".join(from_packed.serialize())) == want
        third = len(seqs) // 3                                   # the two entry points taking turns on one index
        mixed.insert_sequences_packed(seqs[:third])
        mixed.insert_sequences(seqs[third:2 * third])
        mixed.insert_sequences_packed(seqs[2 * third:])
        assert masked(b"".join(mixed.serialize())) == want
        qs = [seqs[-2][100:1400], seqs[-6][:900], seqs[-4][150:1150], seqs[10]]
        np.testing.assert_array_equal(from_packed.query_sequences(qs), o.query_sequences(qs))
    finally:
        from_chars.close(); from_packed.close(); mixed.close()


def test_packed_synthetic_and_character_builds_agree(hip):
    """The device generator emits the packed form directly; the same genomes as characters through
    mk_index_append and packed on the host through mk_index_append_packed give the same index, and
    all three equal the oracle's.  5 Mb genomes at h = 20 take the 32-bit key path of the reduce
    kernel, a 17 Mb one the 64-bit one, a 12 Mb one 32-bit keys with the top position bit in use."""
    from oracle import oracle as orc
    k, h = 31, 20
    L_ = 1_200_000
    ids = [3, 4, 5]
    seqs = [synth.genome_bases(g, 0, L_) for g in ids]
    o = orc.OracleMiekki(k, h, 8, 33, 200)
    o.insert_sequences(seqs)
    want = masked(o.serialize().tobytes())
    a = hip.Miekki(k, h, 8, 33, 200); b = hip.Miekki(k, h, 8, 33, 200); c = hip.Miekki(k, h, 8, 33, 200)
    try:
        a.insert_synthetic(3, 3, L_)
        b.insert_sequences(seqs)
        c.insert_sequences_packed(seqs)
        for ix in (a, b, c):
            assert masked(b"".join(ix.serialize())) == want
    finally:
        a.close(); b.close(); c.close()
    long_ = synth.genome_bases(77, 0, 17_000_000)                 # >= 2^24 positions: 64-bit keys
    mid_ = synth.genome_bases(78, 0, 12_000_000)                  # 32-bit keys whose position uses all 24 bits
    for batch in ([long_, seqs[0]], [mid_, seqs[1]]):
        o2 = orc.OracleMiekki(k, 17, 8, 33, 200)
        o2.insert_sequences(batch)
        d = hip.Miekki(k, 17, 8, 33, 200)
        try:
            d.insert_sequences_packed(batch)
            assert masked(b"".join(d.serialize())) == masked(o2.serialize().tobytes())
        finally:
            d.close()


def test_settled_cell_groups_and_an_import_into_a_filled_context(hip):
    """The build's scatter kernel asks a coarse summary of the Bloom filter (one bit per 2048 cells: all taken) and
    flags only the k-mers it does not settle.  With real k-mers the last cells of a group fill late (the canonical
    k-mer is the smaller of two), so the settled regime is reached here by IMPORTING a filter whose 2049 reachable cells
    (k = 23, -b 32) are all taken: the appends after it must leave the filter alone and still store the right
    fingerprints.  Then the same context takes an import of a nearly empty filter -- the summary of the filled one must
    be forgotten, or the next appends would skip k-mers that now have work to do.  Both against the oracle, which
    deserialises the same streams and inserts the same genomes."""
    from oracle import oracle as orc
    from miekki_amd import lib as L
    k, h, b = 23, 12, 32
    P, hdr, nb = 1 << h, 39, (1 << b) // 8

    def import_stream(ix, raw, G):
        lib = ix._lib
        L.check(lib.mk_index_import_begin(ix._h, G))
        L.check(lib.mk_index_import_columns(ix._h, 0, P, raw[hdr:hdr + P * G]))
        at = hdr + P * G
        gs = np.frombuffer(raw[at:at + 8 * G], np.uint64).copy(); at += 8 * G
        L.check(lib.mk_index_import_bloom(ix._h, 0, nb, raw[at:at + nb])); at += nb
        ss = np.frombuffer(raw[at:at + 4 * G], np.uint32).copy()
        L.check(lib.mk_index_import_sizes(ix._h, gs.ctypes.data, ss.ctypes.data))
        ix.file_names = [f"g{i}" for i in range(G)]

    seed = [synth.genome_bases(8800 + g, 0, 40_000) for g in range(3)]
    o = orc.OracleMiekki(k, h, 8, b, 200)
    o.insert_sequences(seed)
    raw = np.frombuffer(o.serialize().tobytes(), np.uint8).copy()
    cells0 = hdr + P * 3 + 8 * 3
    raw[cells0:cells0 + 2049] = np.where(raw[cells0:cells0 + 2049] == 0, 1, raw[cells0:cells0 + 2049])   # every reachable cell taken
    full = orc.OracleMiekki.deserialize(raw)
    ix = hip.Miekki(k, h, 8, b, 200)
    try:
        import_stream(ix, raw.tobytes(), 3)
        batches = [[synth.genome_bases(8820 + 10 * i + g, 0, 50_000 + 999 * g) for g in range(4)] for i in range(3)]
        for batch in batches:                                    # the first one sees a forgotten summary, the others a full one
            full.insert_sequences(batch)
            ix.insert_sequences_packed(batch)
        assert masked(b"".join(ix.serialize())) == masked(full.serialize().tobytes())
        qs = [batches[2][1][100:1300], batches[0][3][:900], seed[1][5000:6000]]
        np.testing.assert_array_equal(ix.query_sequences(qs), full.query_sequences(qs))
        # now a nearly empty filter into the same context
        small = [synth.genome_bases(8900 + g, 0, 30_000) for g in range(3)]
        o2 = orc.OracleMiekki(k, h, 8, b, 200)
        o2.insert_sequences(small)
        import_stream(ix, o2.serialize().tobytes(), 3)
        more = [synth.genome_bases(8950 + g, 0, 60_000) for g in range(5)]
        o2.insert_sequences(more)
        ix.insert_sequences_packed(more)
        assert masked(b"".join(ix.serialize())) == masked(o2.serialize().tobytes())
        np.testing.assert_array_equal(ix.query_sequences([more[2][:1000], small[0][:1000]]),
                                      o2.query_sequences([more[2][:1000], small[0][:1000]]))
    finally:
        ix.close()
