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


def oracle_sample_check(ix, k, h, fpb, G, L_, queries, via_columns=False):
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
    return sample
