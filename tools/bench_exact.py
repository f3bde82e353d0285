#!/usr/bin/env python3
"""Exact mode (BASELINE config 5: candidate filter on the GPU + exact k-mer set intersection on
the GPU) on a workload that produces hits: G synthetic 5 Mb genomes at -h 20 (at -h 17 such
genomes have genome_size 0 and nothing reaches the verification), Q 1 kb
queries, filter_results(.., 5, 10, threshold) -> per hit genome K7 (mk_exact_load_genome +
mk_exact_query).  Prints one JSON line: candidate-filter rate, K7 set-build and probe rates.

    python tools/bench_exact.py [G=1000] [Q=20000] [verify_genomes=64]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import miekki_amd  # noqa: E402
import synth  # noqa: E402
from miekki_amd import lib as L  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
NV = int(sys.argv[3]) if len(sys.argv) > 3 else 64
Lg, K, H = 5_000_000, 31, 20
lib = L.load_library()
ix = miekki_amd.Miekki(K, H, 8, 33, 200)
ix.reserve(G)
t0 = time.time()
for g0 in range(0, G, 1024):
    ix.insert_synthetic(g0, min(1024, G - g0), Lg)
L.check(lib.mk_sync(ix._h))
t_build = time.time() - t0
qs = [synth.genome_bases(*synth.query_origin(q, G, Lg, 1000), 1000) for q in range(Q)]
t0 = time.time()
hits, _ = ix.query(qs, 5, 10, 200.0)                                    # query_file_exact's filter (Miekki.cpp:741)
t_filter = time.time() - t0
per_genome = {}
for q, row in enumerate(hits):
    for h in row:
        per_genome.setdefault(h.genome, []).append(q)
genomes = sorted(per_genome)[:NV]
t_load = t_query = 0.0
nk_b = nk_a = 0
confirmed = 0
for g in genomes:
    seq = synth.genome_bases(g, 0, Lg)
    ptrs, lens = L.seq_arrays([seq])
    t0 = time.perf_counter()
    L.check(lib.mk_exact_load_genome(ix._h, ptrs, lens, 1))
    t_load += time.perf_counter() - t0
    nk_b += Lg - K + 1
    batch = [qs[q] for q in per_genome[g]]
    qp, ql = L.seq_arrays(batch)
    inter = np.zeros(len(batch), np.uint64); uni = np.zeros(len(batch), np.uint64)
    t0 = time.perf_counter()
    L.check(lib.mk_exact_query(ix._h, qp, ql, len(batch), inter.ctypes.data, uni.ctypes.data))
    t_query += time.perf_counter() - t0
    nk_a += sum(len(s) - K + 1 for s in batch)
    confirmed += int(((inter >= 900) & np.array([q % G == g for q in per_genome[g]])).sum())
    assert all(int(inter[i]) >= 900 for i, q in enumerate(per_genome[g]) if q % G == g)
print(json.dumps({
    "workload": f"exact mode: {G} synthetic 5 Mb genomes, -k 31 -h 20, {Q} x 1 kb queries, {len(genomes)} hit genomes verified",
    "index_build_s": t_build, "candidate_filter_s": t_filter, "candidate_filter_queries_per_s": Q / t_filter,
    "hit_genomes": len(per_genome), "verified_genomes": len(genomes), "confirmed_source_hits": confirmed,
    "k7_load_genome_ms": 1e3 * t_load / max(len(genomes), 1), "k7_set_build_kmers_per_s": nk_b / max(t_load, 1e-9),
    "k7_query_ms_per_genome": 1e3 * t_query / max(len(genomes), 1), "k7_query_kmers_per_s": nk_a / max(t_query, 1e-9),
    "note": "K7 times include the host->device copy of the genome text (5 MB) and of the queries; bound = scattered 64-bit atomicCAS into an HBM hash set"}))
ix.close()
