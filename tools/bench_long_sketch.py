#!/usr/bin/env python3
"""Sketch time of long queries (more than 4,096 k-mers each: the build's packed kernels + one gating launch), the set
prepared afresh for every step.   python tools/bench_long_sketch.py [genomes] [queries] [query_len]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miekki_amd
from miekki_amd import lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
QL = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
lib = L.load_library()
ix = miekki_amd.Miekki(31, 20, 8, 33, 200)
ix.reserve(G)
ix.insert_synthetic(0, G, 5_000_000)
qs = C.c_void_p()
L.check(lib.mk_qset_synthetic(ix._h, 0, Q, G, 5_000_000, QL, C.byref(qs)))
cap = 128
d_count = torch.zeros(Q, dtype=torch.int32, device="cuda")
d_cand = torch.zeros(Q * cap * 24, dtype=torch.uint8, device="cuda")
def step():
    L.check(lib.mk_qset_invalidate(ix._h, qs))
    L.check(lib.mk_qset_run(ix._h, qs, 10, 10, 100.0, cap, d_count.data_ptr(), d_cand.data_ptr()))
    L.check(lib.mk_sync(ix._h))
step(); ix.reset_stats()
t = time.perf_counter(); step(); dt = time.perf_counter() - t
st = ix.stats()
print(json.dumps({"workload": f"{Q} x {QL} b queries vs {G} genomes, -h 20", "s_per_step": dt, "sketch_ms": st["sketch_ms"],
                  "scan_ms": st["scan_ms"], "with_candidates": int((d_count > 0).sum())}))
lib.mk_qset_free(ix._h, qs); ix.close()
