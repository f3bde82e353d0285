#!/usr/bin/env python3
"""PCIe-fed build rate on its own (the `sketch.host_fed_*` figures of bench.py, more rounds): 64 synthetic 5 Mb genomes
in page-locked host buffers, appended `rounds` times packed (mk_index_append_packed) and as characters (mk_index_append).
    python tools/host_fed_rate.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from miekki_amd import lib as L

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = L.load_library()
for packed in (True, False):
    r = bench.host_fed_build_rate(lib, L, 0, 20, 8, packed, 64, rounds)
    per = 1.25e6 if packed else 5e6
    print(f"{'packed' if packed else 'characters'}: {r:8.0f} sketches/s = {r * per / 1e9:5.1f} GB/s over PCIe ({rounds} batches of 64)")
