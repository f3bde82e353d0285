"""Checks shared by the GPU test modules (not a test module itself)."""
import hashlib
import struct

import numpy as np

import synth


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def export_genomes(ix, ids):
    """Columns of the genomes `ids` through mk_index_export_genomes -> uint8 [P, len(ids), W] (dump byte order)."""
    from miekki_amd import lib as L
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.empty((ix.number_minimizer, len(ids), ix.W), np.uint8)
    L.check(ix._lib.mk_index_export_genomes(ix._h, ids.ctypes.data, len(ids), out.ctypes.data))
    return out


def oracle_sample_check(ix, k, h, fpb, G, L_, queries, via_columns=False, with_oracle=False):
    """Full-size collections cannot be rebuilt in the oracle, single genomes can: for eight genomes spread over
    the id range, the column the HIP index holds (every partition, exported through mk_index_export_genomes -- or,
    via_columns, cut out of a full mk_index_export_columns pass, which costs the whole matrix over PCIe)
    must be the oracle's sketch of the same synthetic genome, byte for byte; and the dense score rows of
    `queries` restricted to those genomes must be what the oracle's query_sequences gives over exactly those
    eight columns UNDER THE COLLECTION'S OWN BLOOM FILTER (exported from the index: the gate depends on all
    genomes, the scores of a column only on the column and the gate)."""
    from oracle import oracle as orc
    from miekki_amd import lib as L
    P, W = 1 << h, fpb // 8
    sample = sorted({0, 1, G // 3, G // 2, G // 2 + 1, (2 * G) // 3, G - 2, G - 1})
    o = orc.OracleMiekki(k, h, fpb, 33, 200)
    want = np.empty((P, len(sample), W), np.uint8)
    for j, g in enumerate(sample):
        fp, _, _ = o.minhash_sketch_partition(synth.genome_bases(g, 0, L_))
        if W == 1:
            want[:, j, 0] = fp.astype(np.uint8)
        else:                                                       # big-endian pairs, as add_index stores them (Miekki.cpp:230-231)
            want[:, j, 0] = (fp >> 8).astype(np.uint8); want[:, j, 1] = (fp & 0xff).astype(np.uint8)
    if via_columns:
        got = np.empty_like(want)
        rows = max(1, min(P, (256 << 20) // (G * W)))
        buf = np.empty(rows * G * W, np.uint8)
        for p0 in range(0, P, rows):
            r = min(rows, P - p0)
            L.check(ix._lib.mk_index_export_columns(ix._h, p0, p0 + r, buf.ctypes.data))
            got[p0:p0 + r] = buf[:r * G * W].reshape(r, G, W)[:, sample, :]
    else:
        got = export_genomes(ix, sample)
    for j, g in enumerate(sample):
        assert sha(got[:, j, :].tobytes()) == sha(want[:, j, :].tobytes()), f"column of genome {g}"
    # an oracle index of just those columns + the collection's Bloom filter
    ss, gs = ix.sketch_size[sample], ix.genome_size[sample]
    nb = ix.bloom_size // 8
    bloom = np.empty(nb, np.uint8)
    L.check(ix._lib.mk_index_export_bloom(ix._h, 0, nb, bloom.ctypes.data))
    hdr = struct.pack("<6IQBBIB", k, h, fpb, 5, len(sample), 33, ix.bloom_size, 0, 0, 200, 1)
    stream = np.concatenate([np.frombuffer(hdr, np.uint8), want.reshape(-1), gs.astype(np.uint64).view(np.uint8), bloom,
                             ss.astype(np.uint32).view(np.uint8)])
    o8 = orc.OracleMiekki.deserialize(stream)
    np.testing.assert_array_equal(ix.query_sequences(queries)[:, sample], o8.query_sequences(queries))
    if with_oracle:
        return sample, o8
    return sample


def oracle_slab_check(ix, o8, sample, L_, filler, nres=10, min_score=10, min_inter=100.0, cap=128):
    """An oracle number next to a slab-path number at full size (Miekki.cpp:344-397): queries cut from the eight sample
    genomes go INSIDE a set of len(filler) + 16 >= 512 queries through mk_qset_run -- the slab schedule, asserted by the
    context's counters -- and
      * every hit of theirs on a sample genome must be the oracle's (genome, matches, jaccard, intersection) for that
        column, bit for bit (o8 = the oracle over exactly those eight columns under the collection's Bloom filter:
        filter_results' arithmetic per genome needs only the score and the two sizes), the source genome among them on top;
      * no sample genome the oracle's filter keeps with an intersection above the query's weakest hit may be missing;
      * mk_qset_scores' dense rows of the same prepared set equal the oracle's on the sample columns."""
    import ctypes as C
    import torch
    from miekki_amd import distributed as mkd
    from miekki_amd import lib as L
    lib = L.load_library()
    assert len(filler) + 16 >= 512
    mine = []
    for j, g in enumerate(sample):                                   # two per sample genome, anywhere in it
        mine.append((g, synth.genome_bases(g, 1000 + 611_953 * j, 1000)))
        mine.append((g, synth.genome_bases(g, L_ - 1000 - 7919 * j, 1000)))
    qs = list(filler)
    where = []
    step = len(qs) // len(mine)
    for j, (g, s) in enumerate(mine):                                # spread over the set: first and last places, both sides of 256 / 512
        at = {0: 0, 1: 255 + 0, 2: 256 + 2, 3: 511 + 3}.get(j, j * step + j)
        qs.insert(at, s)
    for g, s in mine:
        where.append(qs.index(s))
    want_rows = o8.query_sequences([s for _, s in mine])
    ptrs, lens = L.seq_arrays(qs)
    qset = C.c_void_p()
    L.check(lib.mk_qset_upload(ix._h, ptrs, lens, len(qs), C.byref(qset)))
    try:
        d_count = torch.zeros(len(qs), dtype=torch.int32, device="cuda")
        d_cand = torch.zeros(len(qs) * cap * 24, dtype=torch.uint8, device="cuda")
        before = ix.stats()
        L.check(lib.mk_qset_run(ix._h, qset, nres, min_score, min_inter, cap, d_count.data_ptr(), d_cand.data_ptr()))
        L.check(lib.mk_sync(ix._h))
        after = ix.stats()
        assert after["scan_slab_launches"] > before["scan_slab_launches"]
        assert after["scan_launches"] - before["scan_launches"] == after["scan_slab_launches"] - before["scan_slab_launches"]
        hits, over = mkd.merge_candidates(d_count.cpu().numpy()[None], d_cand.cpu().numpy()[None], cap, nres)
        assert not over.any()
        col = {g: j for j, g in enumerate(sample)}
        compared = 0
        for (g, s), at, row in zip(mine, where, want_rows):
            got = hits[at]
            want = {sample[w[0]]: w for w in o8.filter_results(row, len(sample), min_score, min_inter)}
            assert len(got) and int(got[0]["genome"]) == g, (g, at)
            assert g in want
            for x in got:
                gg = int(x["genome"])
                if gg in col:
                    w = want[gg]
                    assert (int(x["matches"]), float(x["jaccard"]), float(x["intersection"])) == (w[1], w[2], w[3]), (g, gg)
                    compared += 1
            weakest = float(got[-1]["intersection"]) if len(got) == nres else -1.0
            for gg, w in want.items():
                if w[3] > weakest:
                    assert gg in {int(x["genome"]) for x in got}, (g, gg)
        assert compared >= len(mine)
        d_scores = torch.zeros(ix.index_size, dtype=torch.int32, device="cuda")
        for at, row in zip(where, want_rows):
            L.check(lib.mk_qset_scores(ix._h, qset, at, at + 1, d_scores.data_ptr()))
            L.check(lib.mk_sync(ix._h))
            np.testing.assert_array_equal(d_scores.cpu().numpy().view(np.uint32)[sample], row)
    finally:
        lib.mk_qset_free(ix._h, qset)
    return len(mine)
