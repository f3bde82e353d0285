"""Pin the CPU oracle (oracle/) to the REAL reference.

Every fixture under tests/golden/ is an output of the unmodified reference
compiled in the authoring container (tests/golden/make_golden.py).  The oracle
must reproduce all of them: integers bit-exactly, doubles to 1e-12 relative
(the reference is built -Ofast; the north-star tolerance is 1e-6).
"""
import hashlib
import os

import numpy as np
import pytest

import synth
from oracle import oracle as orc

CASE_NAMES = ["messy", "h20", "w16", "c1", "c2mini", "h16z", "rnd0", "rnd1", "rnd2", "rnd3", "rnd4", "rnd5"]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def built():
    """case name -> (case, golden npz, oracle index with all genomes inserted)"""
    cache = {}

    def get(name):
        if name not in cache:
            case = synth.CASES[name]()
            gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
            ix = orc.OracleMiekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
            seqs = case.genome_sequences()
            for i in range(0, len(seqs), 11):                 # flush every 11 like Miekki.cpp:571
                ix.insert_sequences(seqs[i:i + 11])
            cache[name] = (case, gold, ix)
        return cache[name]
    return get


@pytest.mark.parametrize("name", CASE_NAMES)
def test_index_build_matches_reference(built, name):
    case, gold, ix = built(name)
    assert ix.index_size == int(gold["G"])
    np.testing.assert_array_equal(ix.sketch_size, gold["sketch_size"])
    np.testing.assert_array_equal(ix.genome_size, gold["genome_size"])
    cols = ix.columns()
    W = ix.W
    np.testing.assert_array_equal(cols[:64], gold["cols_head"])
    np.testing.assert_array_equal(cols[-64:], gold["cols_tail"])
    for g in range(ix.index_size):
        assert sha(np.ascontiguousarray(cols[:, g * W:(g + 1) * W]).tobytes()) == str(gold["col_sha_per_genome"][g])
    bloom = ix.bloom
    assert int(np.count_nonzero(bloom)) == int(gold["bloom_nonzero"])
    assert sha(bloom.tobytes()) == str(gold["bloom_sha"])


@pytest.mark.parametrize("name", CASE_NAMES)
def test_index_stream_matches_reference_dump(built, name):
    case, gold, ix = built(name)
    raw = ix.serialize()
    assert raw.size == int(gold["stream_len"])
    raw[32] = 0; raw[38] = 0           # uninitialised byte / compressed flag
    assert sha(raw.tobytes()) == str(gold["stream_sha_masked"])
    back = orc.OracleMiekki.deserialize(ix.serialize())
    assert back.index_size == ix.index_size and back.threshold == ix.threshold
    np.testing.assert_array_equal(back.columns(), ix.columns())
    np.testing.assert_array_equal(back.sketch_size, ix.sketch_size)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_raw_query_sketches(built, name):
    case, gold, ix = built(name)
    qs = case.query_sequences()
    for i in range(len(gold["sk_active"])):
        fp, hs, act = ix.minhash_sketch_partition(qs[i][1])
        assert act == int(gold["sk_active"][i])
        assert sha(fp.tobytes()) == str(gold["sk_fp_sha"][i])
        assert sha(hs.tobytes()) == str(gold["sk_hash_sha"][i])


@pytest.mark.parametrize("name", CASE_NAMES)
def test_scores_bit_exact(built, name):
    case, gold, ix = built(name)
    qs = [s for _, s in case.query_sequences()]
    scores = ix.query_sequences(qs)
    np.testing.assert_array_equal(scores, gold["scores"])
    for q in (0, len(qs) // 2, len(qs) - 1):
        row, act = ix.query_sequence(qs[q])
        np.testing.assert_array_equal(row, gold["scores"][q])
        assert act == int(gold["qseq_active"][q])


PARAMS = {"approx": lambda t: (10, 10, 0.5 * t), "exact_a": lambda t: (5, 10, float(t)),
          "exact_A": lambda t: (5, 5, float(t)), "loose": lambda t: (3, 0, 0.0),
          "loose10": lambda t: (10, 1, 0.0)}


@pytest.mark.parametrize("name", CASE_NAMES)
@pytest.mark.parametrize("tag", list(PARAMS))
def test_filter_results(built, name, tag):
    case, gold, ix = built(name)
    nres, ms, mi = PARAMS[tag](case.threshold)
    off = gold[f"hits_{tag}_off"]
    for q in range(int(gold["nq"])):
        hits = ix.filter_results(gold["scores"][q], nres, ms, mi)
        lo, hi = int(off[q]), int(off[q + 1])
        assert [h[0] for h in hits] == list(gold[f"hits_{tag}_genome"][lo:hi]), (q, tag)
        assert [h[1] for h in hits] == list(gold[f"hits_{tag}_matches"][lo:hi])
        np.testing.assert_allclose([h[2] for h in hits], gold[f"hits_{tag}_jaccard"][lo:hi], rtol=1e-12, atol=0)
        np.testing.assert_allclose([h[3] for h in hits], gold[f"hits_{tag}_inter"][lo:hi], rtol=1e-12, atol=0)


def test_filter_heap_ties_synthetic():
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "filter_ties.npz"))
    ix = orc.OracleMiekki(31, 2, 8, 0, 0)
    off = gold["off"]
    for c in range(int(gold["n"])):
        G, nres, ms = (int(x) for x in gold[f"c{c}_par"])
        ix.poke_sizes(gold[f"c{c}_ss"], gold[f"c{c}_gs"])
        hits = ix.filter_results(gold[f"c{c}_sc"], nres, ms, float(gold[f"c{c}_mi"]))
        lo, hi = int(off[c]), int(off[c + 1])
        assert [h[0] for h in hits] == list(gold["genome"][lo:hi]), c
        assert [h[1] for h in hits] == list(gold["matches"][lo:hi]), c
        np.testing.assert_allclose([h[3] for h in hits], gold["inter"][lo:hi], rtol=1e-12)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_out_txt_matches_reference_cli(built, name, golden_dir):
    """-a output, Miekki.cpp:426-483 at -t 1: one line per kept record, in file order."""
    case, gold, ix = built(name)
    want = open(os.path.join(golden_dir, f"{name}_out.txt"), "rb").read()
    nres, ms, mi = PARAMS["approx"](case.threshold)
    got = b""
    for q, (hd, sq) in enumerate(case.query_sequences()):
        got += ix.format_query_line(hd, ix.filter_results(gold["scores"][q], nres, ms, mi))
    assert got == want


@pytest.mark.parametrize("name", CASE_NAMES)
def test_exact_mode_matches_reference_cli(built, name, golden_dir):
    """-e output (Miekki.cpp:723-859).  Lines are compared as a multiset of
    (real_jax, jaccard_est, nb_inter, inter_est, header, file): the reference
    groups them per genome file through an unordered_map."""
    case, gold, ix = built(name)
    want = open(os.path.join(golden_dir, f"{name}_exact.txt"), "rb").read().decode().splitlines()
    files = [(fn, data) for fn, data, _ in case.genome_files
             if len(b"".join(l for l in data.split(b"\n") if not l.startswith(b">"))) >= case.k]
    sets = {}
    got = []
    nres, ms, mi = PARAMS["exact_a"](case.threshold)
    q = -1
    for hd, sq in case.query_sequences():
        q += 1
        if sq[:1] not in (b"A", b"C", b"G", b"T", b"N"):          # Miekki.cpp:736
            continue
        for g, m, jac, inter in ix.filter_results(gold["scores"][q], nres, ms, mi):
            fn, data = files[g]
            if fn not in sets:
                sets[fn] = orc.exact_genome_set(data, case.k)
            ni, nu = orc.exact_query(sets[fn], sq, case.k)
            if ni > 0:
                got.append("%g\t%g\t%g\t%g\t%s\t%s" % (ni / nu, jac, ni, inter, hd.decode(), fn))
    assert sorted(got) == sorted(want)


def test_config2_regime_is_pinned(built):
    """BASELINE config 2's regime (5 Mb genomes at -h 17): full sketches, active^2 wraps to 0 in the
    reference's u32 arithmetic (Miekki.cpp:289, 306), genome_size 0 -- so the reference reports the
    right scores and no hits.  The fixture is the real reference's output; the oracle reproduces it."""
    case, gold, ix = built("c2mini")
    assert (gold["sketch_size"] == 1 << 17).all() and (gold["genome_size"] == 0).all()
    assert int(gold["hits_approx_off"][-1]) == 0                        # not a single hit in approximate mode
    sc = gold["scores"]
    for q in range(30):                                                  # ... although the source genome stands out
        assert sc[q, q % 3] >= 10 and sc[q, q % 3] > 5 * np.delete(sc[q], q % 3).max()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "c2mini_out.txt"), "rb").read().splitlines()
    assert all(l.endswith(b":") for l in text) and len(text) == int(gold["nq"])
    _, gz, _ = built("h16z")
    assert (gz["sketch_size"][:2] == 1 << 16).all() and (gz["genome_size"][:2] == 0).all() and (gz["genome_size"][2:] > 0).all()


@pytest.mark.parametrize("name", synth.REF_INDEX_CASES)
def test_reference_written_index_file_deserialises(built, name):
    """The reference's own `-d` file (bytes of its zstr writer) -> oracle object -> same stream."""
    import gzip
    case, gold, ix = built(name)
    raw = bytearray(gzip.decompress(open(os.path.join(os.path.dirname(__file__), "golden", f"{name}_ref_idx.gz"), "rb").read()))
    assert len(raw) == int(gold["stream_len"])
    raw[32] = 0; raw[38] = 0
    assert sha(raw) == str(gold["stream_sha_masked"])
    o = orc.OracleMiekki.deserialize(np.frombuffer(bytes(raw), np.uint8))
    again = bytearray(o.serialize().tobytes()); again[32] = 0; again[38] = 0
    assert again == raw
    np.testing.assert_array_equal(o.sketch_size, ix.sketch_size)


def test_tie_heavy_collection_matches_reference_cli(golden_dir):
    """`dups`: 300 copies of one genome -- every copy ties, ties replace in the reference's heap, so the
    printed ten are decided by its sift sequence alone.  The oracle must print the reference's lines."""
    case = synth.EXTRA_CASES["dups"]()
    ix = orc.OracleMiekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
    seqs = case.genome_sequences()
    for i in range(0, len(seqs), 11):
        ix.insert_sequences(seqs[i:i + 11])
    recs = case.query_sequences()
    scores = ix.query_sequences([s for _, s in recs])
    got = b"".join(ix.format_query_line(hd, ix.filter_results(scores[q], 10, 10, 0.5 * case.threshold))
                   for q, (hd, _) in enumerate(recs))
    assert got == open(os.path.join(golden_dir, "dups_out.txt"), "rb").read()


@pytest.mark.parametrize("name", ["h16z", "h20", "w16", "messy"])
def test_insert_sequence_matches_reference_index_file(name):
    """The one-genome path (Miekki.cpp:243-273 behind index_file 518-536): the reference ran every genome file of the
    case through index_file (tests/golden/make_golden.py single:<case>); its size estimate keeps the active count in a
    double -- at -h 16 a full sketch gives 65536^2 = 2^32, which insert_sequences' u32 wraps to 0."""
    case = synth.CASES[name]()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}_single.npz"))
    ix = orc.OracleMiekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
    for s in case.genome_sequences():
        ix.insert_sequence(s)
    assert ix.index_size == int(gold["G"])
    np.testing.assert_array_equal(ix.sketch_size, gold["sketch_size"])
    np.testing.assert_array_equal(ix.genome_size, gold["genome_size"])
    raw = ix.serialize()
    assert raw.size == int(gold["stream_len"])
    raw[32] = 0; raw[38] = 0
    assert sha(raw.tobytes()) == str(gold["stream_sha_masked"])
    if name == "h16z":
        batch = np.load(os.path.join(os.path.dirname(__file__), "golden", "h16z.npz"))
        assert (batch["genome_size"][:2] == 0).all() and (gold["genome_size"][:2] > 0).all()
