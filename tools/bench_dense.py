#!/usr/bin/env python3
"""Whole-genome (-A style) queries: dense path throughput.  Not the headline bench;
records the N3 "next row" measurement.
    python tools/bench_dense.py [G] [n_queries] [h] [fp_bits]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import miekki_amd
from miekki_amd import lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = int(sys.argv[3]) if len(sys.argv) > 3 else 20
FPB = int(sys.argv[4]) if len(sys.argv) > 4 else 8
LEN = 5_000_000
lib = L.load_library()
ix = miekki_amd.Miekki(31, h, FPB, 33, 200)
ix.reserve(G)
ix.insert_synthetic(0, G, LEN)
qs = C.c_void_p()
L.check(lib.mk_qset_synthetic(ix._h, 0, Q, G, LEN + 1, LEN, C.byref(qs)))     # query q = genome q mod G, whole
cap = 128
d_count = torch.zeros(Q, dtype=torch.int32, device="cuda")
d_cand = torch.zeros(Q * cap * 24, dtype=torch.uint8, device="cuda")
def step():
    L.check(lib.mk_qset_run(ix._h, qs, 10, 10, 100.0, cap, d_count.data_ptr(), d_cand.data_ptr()))
    L.check(lib.mk_sync(ix._h))
step(); ix.reset_stats()
t0 = time.perf_counter(); steps = 3
for _ in range(steps): step()
dt = (time.perf_counter() - t0) / steps
act = np.zeros(Q, np.uint32); L.check(lib.mk_qset_active(ix._h, qs, act.ctypes.data))
st = ix.stats()
cmp_ = int(act.sum()) * G
cand = np.frombuffer(d_cand.cpu().numpy().tobytes(), dtype=[("g", "<u4"), ("m", "<u4"), ("j", "<f8"), ("i", "<f8")]).reshape(Q, cap)
top_ok = sum(1 for q in range(Q) if d_count[q].item() and max(cand[q][:min(int(d_count[q].item()), cap)], key=lambda r: r["i"])["g"] == q % G)
print(json.dumps({"workload": f"{Q} whole 5 Mb genomes as queries vs {G} genomes, -h {h}, {FPB}-bit fingerprints", "s_per_step": dt,
                  "comparisons_per_s": cmp_ / dt, "active_per_query": float(act.mean()),
                  "matrix_bytes_read_if_once_per_group": (1 << h) * G * ((Q + 3) // 4),
                  "scan_ms_per_step": st["scan_ms"] / steps, "sketch_ms_per_step": st["sketch_ms"] / steps,
                  "select_ms_per_step": st["filter_ms"] / steps, "top_hit_is_self": top_ok}))
