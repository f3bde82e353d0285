"""Edge cases of the C ABI on the GPU: empty and degenerate inputs, error returns,
matrix re-layout on growth, append after import."""
import ctypes as C
import gzip

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import miekki_amd
    return miekki_amd


def stream_of(ix):
    return b"".join(ix.serialize())


def test_empty_index_and_empty_batches(hip, tmp_path):
    from oracle import oracle as orc
    ix = hip.Miekki(31, 10, 8, 32, 50)
    try:
        assert ix.index_size == 0
        ix.insert_sequences([])
        hits, act = ix.query([synth.genome_bases(0, 0, 500)], 10, 10, 1.0)
        assert hits == [[]] and int(act[0]) == 0
        assert ix.query_sequences([synth.genome_bases(0, 0, 500)]).shape == (1, 0)
        hits, act = ix.query([], 10, 10, 1.0)
        assert hits == []
        # an empty index dumps and loads like the reference's would (header + Bloom only)
        o = orc.OracleMiekki(31, 10, 8, 32, 50)
        raw = bytearray(stream_of(ix)); want = bytearray(o.serialize().tobytes())
        raw[32] = want[32] = 0
        assert bytes(raw) == bytes(want)
        ix.dump_disk(str(tmp_path / "empty.gz"))
        back = hip.Miekki.load(str(tmp_path / "empty.gz"))
        try:
            assert back.index_size == 0 and back.threshold == 50
            back.insert_sequences([synth.genome_bases(1, 0, 4000)])      # append after load
            assert back.index_size == 1
        finally:
            back.close()
    finally:
        ix.close()


def test_queries_without_kmers(hip):
    ix = hip.Miekki(31, 10, 8, 32, 0)
    try:
        g = synth.genome_bases(3, 0, 5000)
        ix.insert_sequences([g])
        qs = [g[:10], g[:31], b"", g[:32], g[100:400]]               # < k, = k (no k-mer processed), empty, one k-mer
        scores = ix.query_sequences(qs)
        assert scores[:3].sum() == 0
        hits, act = ix.query(qs, 10, 0, 0.0)
        assert [int(a) for a in act[:3]] == [0, 0, 0] and int(act[4]) > 0
        assert hits[4] and hits[4][0].genome == 0
    finally:
        ix.close()


def test_error_returns(hip):
    from miekki_amd import lib as L
    lib = L.load_library()
    ix = hip.Miekki(31, 10, 8, 32, 0)
    try:
        with pytest.raises(L.MiekkiHipError) as e:
            ix.insert_sequences([b"ACGT"])                              # shorter than k: the driver's job (Miekki.cpp:569)
        assert e.value.status == -1 and "shorter than k" in str(e.value)
        buf = np.zeros(16, np.uint8)
        assert lib.mk_index_export_columns(ix._h, 5, 2000, buf.ctypes.data) == -1     # partition range out of bounds
        assert lib.mk_index_export_bloom(ix._h, 0, (1 << 29) + 1, buf.ctypes.data) == -1
        assert lib.mk_qset_run(ix._h, None, 10, 10, 1.0, 16, None, None) == -1
    finally:
        ix.close()
    for bad in (dict(k=40), dict(k=1), dict(h=0), dict(h=29), dict(b=31), dict(b=41)):
        p = L.Params(bad.get("k", 31), bad.get("h", 14), 8, bad.get("b", 33), 200, 0, 0, 0)
        h = C.c_void_p()
        assert lib.mk_create(C.byref(p), C.byref(h)) == -1, bad


def test_growth_past_the_reservation_relays_the_matrix(hip):
    """1,100 one-byte genomes exceed the first 1 KiB row pitch: the matrix is re-laid
    out mid-build and must still equal the oracle's."""
    from oracle import oracle as orc
    k, h = 21, 8
    seqs = [synth.genome_bases(8000 + g, 0, 600 + (g % 50)) for g in range(1100)]
    o = orc.OracleMiekki(k, h, 8, 32, 0)
    o.insert_sequences(seqs)
    ix = hip.Miekki(k, h, 8, 32, 0)
    try:
        for i in range(0, len(seqs), 257):
            ix.insert_sequences(seqs[i:i + 257])
        raw = bytearray(stream_of(ix)); want = bytearray(o.serialize().tobytes())
        raw[32] = want[32] = 0
        assert bytes(raw) == bytes(want)
        q = [seqs[1050][50:500], seqs[3]]
        np.testing.assert_array_equal(ix.query_sequences(q), o.query_sequences(q))
    finally:
        ix.close()


def test_append_after_import_equals_one_build(hip, tmp_path):
    k, h = 31, 12
    seqs = [synth.genome_bases(8500 + g, 0, 20_000) for g in range(20)]
    a = hip.Miekki(k, h, 16, 33, 10)
    b = hip.Miekki(k, h, 16, 33, 10)
    try:
        a.insert_sequences(seqs)
        b.insert_sequences(seqs[:9])
        b.dump_disk(str(tmp_path / "part.gz"))
        c = hip.Miekki.load(str(tmp_path / "part.gz"))
        try:
            c.insert_sequences(seqs[9:])
            assert stream_of(c) == stream_of(a)
        finally:
            c.close()
    finally:
        a.close(); b.close()


def test_repetitive_sequences_overflow_the_bins_gracefully(hip):
    """Low-complexity genomes put thousands of identical k-mers into one partition:
    the binned sketch's fixed slots overflow (overflow list, then the atomic-kernel
    fallback when even that is full) and the result must not change."""
    from oracle import oracle as orc
    k, h = 31, 16
    seqs = [b"A" * 300_000,                                             # one k-mer, 300k times
            b"ACGT" * 400_000,                                           # 4 distinct k-mers, 1.6 M positions (> overflow list)
            (synth.genome_bases(1, 0, 977) * 300)[:250_000],             # period-977 repeat
            synth.genome_bases(2, 0, 100_000),                           # a normal one in the same batch
            b"N" * 5000 + synth.genome_bases(3, 0, 50_000) + b"n" * 4000]
    o = orc.OracleMiekki(k, h, 8, 33, 5)
    o.insert_sequences(seqs)
    ix = hip.Miekki(k, h, 8, 33, 5)
    try:
        ix.insert_sequences(seqs)
        np.testing.assert_array_equal(ix.sketch_size, o.sketch_size)
        np.testing.assert_array_equal(ix.genome_size, o.genome_size)
        raw = bytearray(stream_of(ix)); want = bytearray(o.serialize().tobytes())
        raw[32] = want[32] = 0
        assert bytes(raw) == bytes(want)
        qs = [b"A" * 500, b"ACGT" * 300, seqs[2][100:1500], seqs[3][5:1005], seqs[4][4990:6000]]
        np.testing.assert_array_equal(ix.query_sequences(qs), o.query_sequences(qs))
    finally:
        ix.close()


def test_pipelined_appends_from_pinned_buffers(hip):
    """mk_index_append returns once its copy is done and leaves the kernels in flight: the
    caller may overwrite its buffers at once, and back-to-back appends (two alternating
    device buffers, batches of different sizes, a >64-genome call) must give the index of
    one synchronous build.  Buffers come from mk_host_alloc (pinned: the copy is a DMA)."""
    from miekki_amd import lib as L
    lib = L.load_library()
    k, h = 25, 12
    seqs = [synth.genome_bases(900 + i, 0, 20_000 + 997 * (i % 7)) for i in range(150)]
    ref = hip.Miekki(k, h, 8, 33, 20)
    ix = hip.Miekki(k, h, 8, 33, 20)
    try:
        ref.insert_sequences(seqs)
        want = stream_of(ref)
        cap = max(len(s) for s in seqs)
        slots = []
        for _ in range(70):
            p = C.c_void_p()
            L.check(lib.mk_host_alloc(ix._h, cap, C.byref(p)))
            slots.append(p)
        pos = 0
        for n in (3, 64, 1, 70, 12):                                       # 150 genomes; 70 > one device batch
            ptrs = (C.c_char_p * n)()
            lens = (C.c_uint64 * n)()
            for j in range(n):
                s = seqs[pos + j]
                C.memmove(slots[j], s, len(s))
                ptrs[j] = C.cast(slots[j], C.c_char_p)
                lens[j] = len(s)
            L.check(lib.mk_index_append(ix._h, ptrs, lens, n))
            for j in range(n):                                             # the buffers are ours again
                C.memset(slots[j], ord("T"), cap)
            pos += n
        assert ix.index_size == 150
        assert stream_of(ix) == want
        q = [seqs[5][100:1100], seqs[149][:900]]
        np.testing.assert_array_equal(ix.query_sequences(q), ref.query_sequences(q))
        # ... and that one build is the oracle's (not merely the HIP path agreeing with itself)
        from oracle import oracle as orc
        o = orc.OracleMiekki(k, h, 8, 33, 20)
        o.insert_sequences(seqs)
        got = bytearray(stream_of(ix)); exp = bytearray(o.serialize().tobytes())
        got[32] = exp[32] = 0; got[38] = exp[38] = 0
        assert bytes(got) == bytes(exp)
        np.testing.assert_array_equal(ix.query_sequences(q), o.query_sequences(q))
        for p in slots:
            lib.mk_host_free(ix._h, p)
    finally:
        ref.close(); ix.close()


def test_empty_sketch_genomes_and_min_score_zero(hip):
    """A genome exactly k long stores no fingerprint (the last k-mer is skipped, Miekki.cpp:162):
    sketch_size 0, and with min_score 0 its jaccard is 0/0 = NaN (Miekki.cpp:381-383).  The
    reference's heap then does whatever its comparisons yield; mk_query must reproduce that (it
    takes the host replay), and the device-only mk_qset_run must refuse rather than guess."""
    from oracle import oracle as orc
    from miekki_amd import lib as L
    import torch
    lib = L.load_library()
    k, h = 23, 7
    lens = [20000, 5000, k, 60000, k, 300]
    seqs = [synth.genome_bases(5100 + g, 0, n) for g, n in enumerate(lens)]
    o = orc.OracleMiekki(k, h, 8, 32, 0)
    o.insert_sequences(seqs)
    ix = hip.Miekki(k, h, 8, 32, 0)
    try:
        ix.insert_sequences(seqs)
        assert list(ix.sketch_size) == list(o.sketch_size) and int(ix.sketch_size[2]) == 0
        qs = [seqs[3][:900], seqs[3][:k], seqs[0][:6000], seqs[5]]
        scores = o.query_sequences(qs)
        for nres, ms, mi in ((1, 0, 25.0), (3, 0, 25.0), (1, 0, 0.0), (10, 0, 0.0), (2, 1, 0.0)):
            hits, _ = ix.query(qs, nres, ms, mi)
            for q in range(len(qs)):
                want = o.filter_results(scores[q], nres, ms, mi)
                assert [(x.genome, x.matches) for x in hits[q]] == [(w[0], w[1]) for w in want], (nres, ms, mi, q)
        ptrs, lens_a = L.seq_arrays(qs)
        qset = C.c_void_p()
        L.check(lib.mk_qset_upload(ix._h, ptrs, lens_a, len(qs), C.byref(qset)))
        d_count = torch.zeros(len(qs), dtype=torch.int32, device="cuda")
        d_cand = torch.zeros(len(qs) * 16 * 24, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert lib.mk_qset_run(ix._h, qset, 5, 0, 0.0, 16, d_count.data_ptr(), d_cand.data_ptr()) == -2   # MK_ERR_UNSUPPORTED
        assert b"NaN" in lib.mk_last_error()
        assert lib.mk_qset_run(ix._h, qset, 5, 1, 0.0, 16, d_count.data_ptr(), d_cand.data_ptr()) == 0
        L.check(lib.mk_sync(ix._h))
        lib.mk_qset_free(ix._h, qset)
    finally:
        ix.close()


def test_very_large_query_call_is_sliced(hip):
    """More than 2^18 queries in ONE mk_query call: answered in slices, same hits as smaller calls."""
    k, h = 11, 8
    seqs = [synth.genome_bases(6200 + g, 0, 4000) for g in range(5)]
    ix = hip.Miekki(k, h, 8, 32, 0)
    try:
        ix.insert_sequences(seqs)
        base = [seqs[q % 5][(7 * q) % 3000:(7 * q) % 3000 + 60 + q % 40] for q in range(1000)]
        qs = base * 270                                              # 270,000 queries
        hits, act = ix.query(qs, 3, 1, 0.0)
        small, act_s = ix.query(base, 3, 1, 0.0)
        assert len(hits) == len(qs)
        for q in (0, 1, 999, 1000, 262143, 262144, 262145, 269999):
            assert [(x.genome, x.matches) for x in hits[q]] == [(x.genome, x.matches) for x in small[q % 1000]], q
            assert int(act[q]) == int(act_s[q % 1000])
    finally:
        ix.close()


def test_exact_mode_many_tiny_contigs_and_many_queries(hip):
    """K7 over shapes the first version could not launch: a genome file of 100,000 contigs (most a
    few dozen bases, some shorter than k -- those leak into the next contig, Miekki.cpp:806-812 --
    one of 300 kb, N and lower case inside) and 100,000 queries in ONE call; then a second batch of
    queries against the still-resident genome set.  |A n B| and |A u B| against the oracle."""
    import ctypes as C
    from oracle import oracle as orc
    from miekki_amd import lib as L
    lib = L.load_library()
    rng = np.random.default_rng(2026)
    k = 21
    big = bytearray(synth.genome_bases(4242, 0, 300_000))
    big[1000] = ord("N"); big[5000:5050] = bytes(big[5000:5050]).lower(); big[70_000] = ord("x")
    pool = synth.genome_bases(4243, 0, 4_000_000)
    lens = rng.integers(5, 70, 100_000)
    lens[50_000] = 0                                                     # an empty record
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    parts = []
    for i in range(100_000):
        parts.append(b">c%d\n" % i)
        parts.append(pool[starts[i]:starts[i] + lens[i]] + b"\n")
        if i == 31_337:
            parts.append(b">big\n" + bytes(big) + b"\n")
    fasta = b"".join(parts)
    ix = hip.Miekki(k, 10, 8, 32, 0)
    try:
        qs = []
        for q in range(100_000):
            src = pool if q % 3 else bytes(big)
            n = int(rng.integers(k - 2, 160))
            off = int(rng.integers(0, len(src) - n))
            qs.append(src[off:off + n])
        qs[7] = b"ACGTNNNNNNNNNNNNNNNNNNNNNNNNNACGT"
        qs[8] = qs[9]                                                     # duplicates keep their own sets
        inter, uni = ix.ground_truth_batch(qs, fasta)
        kset = orc.exact_genome_set(fasta, k)
        for q in list(range(0, 100_000, 37)) + [7, 8, 9, 99_999]:
            wi, wu = orc.exact_query(kset, qs[q], k)
            assert (int(inter[q]), int(uni[q])) == (wi, wu), q
        # second batch against the resident genome set (the CLI's flush-at-100 path)
        qs2 = [bytes(big[1000 * i:1000 * i + 500]) for i in range(64)]
        ptrs, lens2 = L.seq_arrays(qs2)
        i2 = np.zeros(64, np.uint64); u2 = np.zeros(64, np.uint64)
        L.check(lib.mk_exact_query(ix._h, ptrs, lens2, 64, i2.ctypes.data, u2.ctypes.data))
        for q in range(64):
            assert (int(i2[q]), int(u2[q])) == orc.exact_query(kset, qs2[q], k), q
        assert int(i2[10]) > 400                                          # really shared
    finally:
        ix.close()


def test_dev_copy_counts_the_path_it_took(hip):
    """mk_dev_copy between two contexts (here both on GPU 0: the same-device path counts as direct) is complete on
    return and is counted in the SOURCE context's stats: a staged exchange must show up as such."""
    import ctypes as C
    from miekki_amd import lib as L
    a = hip.Miekki(21, 10, 8, 32, 10); b = hip.Miekki(21, 10, 8, 32, 10)
    lib = a._lib
    try:
        n = 1 << 20
        src = np.arange(n, dtype=np.uint8)
        da, db = C.c_void_p(), C.c_void_p()
        L.check(lib.mk_dev_alloc(a._h, n, C.byref(da))); L.check(lib.mk_dev_alloc(b._h, n, C.byref(db)))
        L.check(lib.mk_dev_upload(a._h, da, src.ctypes.data, n))
        before = a.stats()
        L.check(lib.mk_dev_copy(b._h, db, a._h, da, n))
        back = np.zeros(n, np.uint8)
        L.check(lib.mk_dev_download(b._h, back.ctypes.data, db, n))
        np.testing.assert_array_equal(back, src)
        after = a.stats()
        assert after["peer_copies"] == before["peer_copies"] + 1 and after["peer_copy_bytes"] == before["peer_copy_bytes"] + n
        assert after["staged_copies"] == before["staged_copies"] == 0
        assert b.stats()["peer_copies"] == 0                      # counted where the copy was queued
        lib.mk_dev_free(a._h, da); lib.mk_dev_free(b._h, db)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("b", [32, 33])
def test_bloom_positions_when_the_low_word_carries(hip, b):
    """universal_hash(anc, i) = canon + (i * 69 * revhash64(anc)) % 1024 (utils.cpp:197-199, called on the HASH): the
    second term matters only when the low word of the canonical k-mer can carry (> 0xFFFFFC00, one k-mer in four million)
    -- no seeded test ever meets one.  Constructed: k = 31, canonical k-mer A <14 free bases> T^15 G / T^15 A / T^15 C (low
    word 0xFFFFFFFE / ...FC / ...FD; the other strand starts with C / T / G: larger), as the ONLY processed k-mer of a
    32-base genome, so it is a partition's winner and goes into the filter.  Index stream (Bloom byte VALUES included) and
    gated scores must be the oracle's; at -b 32 the carry changes the bit (hash % 8), at -b 33 only after a second one."""
    from oracle import oracle as orc
    rng = np.random.default_rng(7 + b)
    genomes = []
    for tail in (b"T" * 15 + b"G", b"T" * 15 + b"A", b"T" * 15 + b"C"):   # (the other strand then starts with C, T, G: larger)
        for _ in range(16):
            mid = bytes(rng.choice(list(b"ACGT"), 14).tolist())
            genomes.append(b"A" + mid + tail + b"A")                 # 32 bases: two k-mers, the last one is skipped (Miekki.cpp:162)
    genomes += [synth.genome_bases(50 + i, 0, 3000) for i in range(3)]
    o = orc.OracleMiekki(31, 12, 8, b, 0)
    o.insert_sequences(genomes)
    ix = hip.Miekki(31, 12, 8, b, 0)
    try:
        ix.insert_sequences(genomes[:20]); ix.insert_sequences_packed(genomes[20:])
        np.testing.assert_array_equal(ix.sketch_size, o.sketch_size)
        assert (o.sketch_size[:48] == 1).sum() >= 40                # (a fingerprint of 255 cannot be stored: one k-mer in sixteen)
        raw = np.frombuffer(stream_of(ix), np.uint8).copy()
        want = o.serialize()
        raw[32] = want[32] = 0
        assert raw.size == want.size and raw.tobytes() == want.tobytes()
        qs = genomes[:48] + [g[:31] + b"C" for g in genomes[:8]]
        np.testing.assert_array_equal(ix.query_sequences(qs), o.query_sequences(qs))
        fresh = hip.Miekki(31, 12, 8, b, 0)                          # the same k-mers as QUERIES against a filter that lacks them
        try:
            fresh.insert_sequences(genomes[48:])
            o2 = orc.OracleMiekki(31, 12, 8, b, 0)
            o2.insert_sequences(genomes[48:])
            np.testing.assert_array_equal(fresh.query_sequences(qs), o2.query_sequences(qs))
        finally:
            fresh.close()
    finally:
        ix.close()


def test_first_batch_on_fresh_contexts_sharing_a_gpu_keeps_its_exceptions(hip):
    """The race fixed in round 5 (build.hip: a build side's counters were cleared on the back stream and could land after the
    front stage had set a batch's exception flags): the FIRST batch of a context whose Bloom arrays have just been allocated
    (-b 33: a gigabyte's fill still queued), genomes with N runs, lower case and junk -- on four contexts that share device 0
    and build at the same time, once from characters (mk_index_append) and once from gzip'd files (mk_gz_unpack +
    mk_index_append_gz).  Every context must hold what the oracle holds."""
    import threading
    import zlib
    from oracle import oracle as orc
    k, h = 31, 12
    rng = np.random.default_rng(99)
    genomes = []
    for g in range(24):
        s = bytearray(synth.genome_bases(500 + g, 0, 150_000 + 1000 * g))
        for _ in range(30):                                          # N runs, lower case, junk: exception bits in most workgroups
            a = int(rng.integers(0, len(s) - 400))
            n = int(rng.integers(1, 300))
            kind = int(rng.integers(0, 3))
            s[a:a + n] = b"N" * n if kind == 0 else bytes(s[a:a + n]).lower() if kind == 1 else bytes(rng.integers(63, 90, n, dtype=np.uint8))   # (no '>': at a line's start it would make the line a header)
        genomes.append(bytes(s))
    want = orc.OracleMiekki(k, h, 8, 33, 10)
    want.insert_sequences(genomes)
    # what is compared: header, columns, genome sizes and the filter's first 64 MiB -- every byte a 62-bit k-mer can reach
    # at -b 33 -- (not the gigabyte of zeros behind them), and the sketch sizes
    head_bytes = 39 + (1 << h) * len(genomes) + 8 * len(genomes) + (64 << 20)
    want_head = bytearray(want.serialize()[:head_bytes].tobytes()); want_head[32] = 0

    def head_of(ix):
        out = bytearray()
        for piece in ix.serialize():
            out += piece
            if len(out) >= head_bytes:
                break
        out[32] = 0
        return bytes(out[:head_bytes])
    blobs = []
    for g, s in enumerate(genomes):
        text = b">g%d\n" % g + b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n"
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        blobs.append(c.compress(text) + c.flush())
    for how in ("chars", "gz"):
        ctxs = [hip.Miekki(k, h, 8, 33, 10) for _ in range(4)]       # fresh: nothing has run on them yet
        errs = []

        def build(ix):
            try:
                if how == "chars":
                    ix.insert_sequences(genomes)
                else:
                    assert ix.insert_gz_files(blobs, fallback=False) == [0] * len(blobs)
            except Exception as e:                                   # noqa: BLE001
                errs.append(e)

        try:
            th = [threading.Thread(target=build, args=(ix,)) for ix in ctxs]
            for t in th: t.start()
            for t in th: t.join()
            assert not errs, errs
            for ix in ctxs:
                np.testing.assert_array_equal(ix.sketch_size, want.sketch_size)
                np.testing.assert_array_equal(ix.genome_size, want.genome_size)
                assert head_of(ix) == bytes(want_head), how
        finally:
            for ix in ctxs: ix.close()
