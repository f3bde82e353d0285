#!/usr/bin/env python3
"""Device-only index build rate (K1-K3 on synthetic genomes), steady state.
    python tools/build_rate.py [genomes] [h]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miekki_amd
from miekki_amd import lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = L.load_library()
ix = miekki_amd.Miekki(31, h, 8, 33, 200)
ix.reserve(G)
ix.insert_synthetic(0, G // 2, 5_000_000); lib.mk_sync(ix._h); ix.reset_stats()      # fills the Bloom filter
t = time.time(); ix.insert_synthetic(G // 2, G - G // 2, 5_000_000); lib.mk_sync(ix._h); dt = time.time() - t
st = ix.stats(); nb = (G - G // 2) / 64
print(f"{(G - G // 2) / dt:.0f} sketches/s; per 64-genome batch: wall {dt / nb * 1e3:.2f} ms, "
      f"sketch {st['build_sketch_ms'] / nb:.2f} ms, finalize+bloom {st['build_finalize_ms'] / nb:.2f} ms")
ix.close()
