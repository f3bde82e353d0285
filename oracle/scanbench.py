"""CPU baseline of bench.py when the reference build (oracle/_ref) is absent: the oracle's
query_sequences timed on the SAME sample shape as `ref_harness scanbench`.

TEST / MEASUREMENT INFRASTRUCTURE -- only bench.py's cpu_baseline leg imports this.  An index
of G genomes is made from the real sketches of four synthetic 5 Mb genomes, columns padded
cyclically (column p, genome g = sketch of genome g % 4 at partition (p + g / 4) mod 2^h),
Bloom filter saturated (the >= 10^4-genome regime), then mko_query_sequences
(Miekki.cpp:344-372 restated) runs over synthetic 1 kb queries on one core.
"""
from __future__ import annotations

import os
import struct
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "tests"))

from . import oracle as orc  # noqa: E402

_HDR = struct.Struct("<6IQBBIB")


def run(h: int, G: int, nq: int, L: int = 5_000_000, qlen: int = 1000, nsrc: int = 4, bloom_log2: int = 33):
    import synth
    k, P = 31, 1 << h
    o = orc.OracleMiekki(k, h, 8, bloom_log2, 200)
    src = np.empty((nsrc, P), np.uint8)
    genomes = []
    for g in range(nsrc):
        s = synth.genome_bases(g, 0, L)
        genomes.append(s)
        fp, _, _ = o.minhash_sketch_partition(s)
        src[g] = fp.astype(np.uint8)
    ss0 = int((src[0] != 255).sum())
    rows = np.empty((G, P), np.uint8)                # genome-major, transposed below
    for g in range(G):
        rows[g] = np.roll(src[g % nsrc], -(g // nsrc))
    cols = np.ascontiguousarray(rows.T)
    del rows
    nb = (1 << bloom_log2) // 8
    stream = np.empty(39 + P * G + 8 * G + nb + 4 * G, np.uint8)
    stream[:39] = np.frombuffer(_HDR.pack(k, h, 8, 5, G, bloom_log2, 1 << bloom_log2, 0, 0, 200, 1), np.uint8)
    at = 39
    stream[at:at + P * G] = cols.reshape(-1); at += P * G
    del cols
    stream[at:at + 8 * G] = np.full(G, L, np.uint64).view(np.uint8); at += 8 * G
    stream[at:at + nb] = 1; at += nb                  # saturated
    stream[at:at + 4 * G] = np.full(G, ss0, np.uint32).view(np.uint8)
    ix = orc.OracleMiekki.deserialize(stream)
    del stream
    qs = [genomes[q % nsrc][off:off + qlen] for q in range(nq)
          for off in [synth.query_origin(q, nsrc, L, qlen)[1]]]
    act = sum(int((ix.minhash_sketch_partition(s)[0] != 255).sum()) for s in qs)
    t0 = time.time()
    chk = 0
    for b in range(0, nq, 201):                       # the reference's batches of 201 (Miekki.cpp:471)
        chk += int(ix.query_sequences(qs[b:b + 201])[:, 0].sum())
    dt = time.time() - t0
    return {"comparisons": act * G, "seconds": dt, "threads": 1, "h": h, "G": G, "queries": nq, "check": chk}


if __name__ == "__main__":
    import json
    print(json.dumps(run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))))
