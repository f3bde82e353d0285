#!/usr/bin/env python3
"""Latency of small mk_query calls (one and 16 queries of 1 kb) against a resident index.
    python tools/latency.py [genomes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth, miekki_amd
G = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
ix = miekki_amd.Miekki(31, 20, 8, 33, 200)
ix.reserve(G); ix.insert_synthetic(0, G, 5_000_000)
qs = [synth.genome_bases(*synth.query_origin(q, G, 5_000_000, 1000), 1000) for q in range(64)]
for n in (1, 16):
    ix.query(qs[:n]); t = time.perf_counter(); reps = 100
    for r in range(reps): hits, _ = ix.query(qs[(r % 4) * n:(r % 4) * n + n])
    dt = (time.perf_counter() - t) / reps
    print(f"{n} queries per call vs {G} genomes: {dt * 1e3:.3f} ms per call; top hit ok: {hits[0][0].genome == ((r % 4) * n) % G}")
ix.close()
