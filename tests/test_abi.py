"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/miekki_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

import miekki_amd
from miekki_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "miekki_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mk_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("mk_create", "mk_index_append", "mk_query", "mk_query_scores", "mk_qset_run",
                 "mk_index_export_columns", "mk_index_import_columns", "mk_exact", "mk_filter_candidates"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(L.library_path())
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/miekki_hip.h but not exported"
    assert sorted(L.SIGNATURES) == declared_symbols()


def test_abi_version_and_struct_layouts():
    lib = L.load_library()
    assert lib.mk_abi_version() == 5
    assert ctypes.sizeof(L.Hit) == 24          # similarity_score, Miekki.h:27-31
    assert ctypes.sizeof(L.Params) == 32
    assert ctypes.sizeof(L.PackedSeq) == 56    # mk_packed_seq: two pointers, the length, 32 head characters
    assert ctypes.sizeof(L.Stats) == 16 * 8    # mk_stats: sixteen 8-byte fields


def test_filter_candidates_is_the_reference_heap(golden_dir):
    """mk_filter_candidates is pure host code: pin it to the reference's tie cases."""
    import numpy as np
    lib = L.load_library()
    gold = np.load(os.path.join(golden_dir, "filter_ties.npz"))
    off = gold["off"]
    for c in range(int(gold["n"])):
        G, nres, ms = (int(x) for x in gold[f"c{c}_par"])
        mi = float(gold[f"c{c}_mi"])
        ss, gs, sc = gold[f"c{c}_ss"], gold[f"c{c}_gs"], gold[f"c{c}_sc"]
        cand = (L.Hit * max(G, 1))()
        n = 0
        for g in range(G):
            if sc[g] < ms:
                continue
            jac = float(sc[g]) / float(ss[g]); inter = jac * float(gs[g])
            if inter < mi:
                continue
            cand[n] = L.Hit(g, int(sc[g]), jac, inter); n += 1
        out = (L.Hit * max(nres, 1))()
        m = lib.mk_filter_candidates(cand, n, nres, out)
        lo, hi = int(off[c]), int(off[c + 1])
        assert [out[i].genome for i in range(m)] == list(gold["genome"][lo:hi]), c
        assert [out[i].matches for i in range(m)] == list(gold["matches"][lo:hi]), c


def test_unsupported_fingerprint_width_fails_like_the_reference():
    # -f values other than 3 / 11: "not implemented" (Miekki.cpp:235-237); checked before any device use
    lib = L.load_library()
    p = L.Params(31, 14, 21, 33, 200, 0, 0, 0)
    h = ctypes.c_void_p()
    assert lib.mk_create(ctypes.byref(p), ctypes.byref(h)) == -2
    assert b"not implemented" in lib.mk_last_error()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.MiekkiHipError):
        miekki_amd.Miekki(31, 14, 8, 33, 200)


def _entrants(vals, nres):
    """Python statement of select.hip: indices that filter_results does not skip
    (Miekki.cpp:386-387), tracking only the multiset of the current top-N."""
    top, out = [], []
    for i, x in enumerate(vals):
        if len(top) >= nres:
            if nres == 0 or min(top) > x:
                continue
            top.remove(min(top))
        top.append(x)
        out.append(i)
    return out


def test_heap_entrants_reproduce_the_full_replay_and_shard_merges():
    """The claim select.hip and the multi-GPU merge rest on: replaying the reference
    heap over the entrants only -- of the whole row, or of contiguous shards
    concatenated in order -- gives the reference's result, ties included."""
    import numpy as np
    lib = L.load_library()
    rng = np.random.default_rng(7)

    def heap(cands, nres):
        buf = (L.Hit * max(len(cands), 1))(*[L.Hit(g, 1, 0.0, float(v)) for g, v in cands])
        out = (L.Hit * max(nres, 1))()
        n = lib.mk_filter_candidates(buf, len(cands), nres, out)
        return [(out[i].genome, out[i].intersection) for i in range(n)]

    for trial in range(300):
        m = int(rng.integers(0, 400))
        nres = int(rng.choice([1, 2, 5, 10, 64]))
        vals = rng.integers(0, int(rng.choice([3, 8, 50, 1000])), m).astype(float).tolist()   # few distinct values: many ties
        full = heap(list(enumerate(vals)), nres)
        ent = _entrants(vals, nres)
        assert heap([(i, vals[i]) for i in ent], nres) == full, trial
        # contiguous shards, each emitting only its own entrants
        cuts = sorted(rng.integers(0, m + 1, int(rng.integers(1, 8))).tolist())
        bounds = [0] + cuts + [m]
        merged = []
        for a, b in zip(bounds, bounds[1:]):
            merged += [(a + i, vals[a + i]) for i in _entrants(vals[a:b], nres)]
        assert heap(merged, nres) == full, trial


def test_python_host_formatting_reproduces_reference_out_txt(golden_dir):
    """Host logic without a GPU: Miekki.format_hits (Miekki.cpp:440-444) over the golden
    hit tuples gives the reference's out.txt byte for byte; _read_text sniffs gzip."""
    import gzip
    import numpy as np
    import synth
    from miekki_amd.index import Miekki, SimilarityScore, _read_text
    for name in ("messy", "h20", "w16", "c1"):
        gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
        case = synth.CASES[name]()
        off = gold["hits_approx_off"]
        text = b""
        for q, (head, _) in enumerate(case.query_sequences()):
            lo, hi = int(off[q]), int(off[q + 1])
            hits = [SimilarityScore(int(gold["hits_approx_genome"][i]), int(gold["hits_approx_matches"][i]),
                                    float(gold["hits_approx_jaccard"][i]), float(gold["hits_approx_inter"][i]))
                    for i in range(lo, hi)]
            text += Miekki.format_hits(head, hits)
        assert text == open(os.path.join(golden_dir, f"{name}_out.txt"), "rb").read(), name


def test_read_text_sniffs_gzip(tmp_path):
    import gzip
    from miekki_amd.index import _read_text
    (tmp_path / "a.fa").write_bytes(b">x\nACGT\n")
    (tmp_path / "b.fa").write_bytes(gzip.compress(b">x\nACGT\n") + gzip.compress(b"TTTT\n"))   # two members
    assert _read_text(str(tmp_path / "a.fa")) == b">x\nACGT\n"
    assert _read_text(str(tmp_path / "b.fa")) == b">x\nACGT\nTTTT\n"
