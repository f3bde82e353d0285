#!/usr/bin/env python3
"""Device-only index build rate (K1-K3 on synthetic genomes): steady state (the second half of the genomes, into a
Bloom filter that the first half has filled), or -- `young` -- a whole collection from an EMPTY index (BASELINE config 2
is such a build from its first genome to its last: 1000 genomes, -h 17).
    python tools/build_rate.py [genomes] [h] [young] [fp_bits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miekki_amd
from miekki_amd import lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
young = len(sys.argv) > 3 and sys.argv[3] == "young"
fpb = int(sys.argv[4]) if len(sys.argv) > 4 else 8
lib = L.load_library()
if young:
    warm = miekki_amd.Miekki(31, h, fpb, 33, 200)          # scratch allocations, code loading: not what is timed
    warm.insert_synthetic(0, 128, 5_000_000); lib.mk_sync(warm._h); warm.close()
ix = miekki_amd.Miekki(31, h, fpb, 33, 200)
ix.reserve(G + 64)
first = 0 if young else G // 2
if young:
    ix.insert_synthetic(10_000_000, 64, 1000); lib.mk_sync(ix._h)       # this context's scratch (64 tiny genomes: ids 0..63 of the index)
    first = 0
else:
    ix.insert_synthetic(0, G // 2, 5_000_000); lib.mk_sync(ix._h)       # fills the Bloom filter
ix.reset_stats()
n = G - first
t = time.time(); ix.insert_synthetic(first, n, 5_000_000); lib.mk_sync(ix._h); dt = time.time() - t
st = ix.stats(); nb = n / 64
print(f"{'young filter: ' if young else ''}{n / dt:.0f} sketches/s ({n} genomes, -h {h}, {fpb}-bit fingerprints); per 64-genome batch: wall {dt / nb * 1e3:.2f} ms, "
      f"sketch {st['build_sketch_ms'] / nb:.2f} ms, finalize+bloom {st['build_finalize_ms'] / nb:.2f} ms")
ix.close()
