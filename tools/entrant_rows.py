#!/usr/bin/env python3
"""How many heap ENTRANTS (Miekki.cpp:387: genomes filter_results does not skip) a shard of G genomes emits per
query -- the occupancy of the 8-byte exchange rows, which decides how wide they must be (entrant_cap in
miekki_amd/shard.py and host/multi_gpu.hpp).  Synthetic genomes / queries of SURVEY.md 8d, query_file's
filter parameters (top 10, min_score 10, min_intersection 100).
    python tools/entrant_rows.py [G ...]      (default 12500 50000 100000)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device buffers)

import miekki_amd  # noqa: E402
from miekki_amd import lib as L  # noqa: E402
from miekki_amd.shard import entrant_cap  # noqa: E402

NQ, CAP = 8192, 1023
lib = L.load_library()
for G in [int(a) for a in sys.argv[1:]] or [12_500, 50_000, 100_000]:
    ix = miekki_amd.Miekki(31, 20, 8, 33, 200)
    ix.reserve(G)
    for g0 in range(0, G, 4096):
        ix.insert_synthetic(g0, min(4096, G - g0), 5_000_000)
    qs = C.c_void_p()
    L.check(lib.mk_qset_synthetic(ix._h, 0, NQ, G, 5_000_000, 1000, C.byref(qs)))
    rows = torch.zeros(NQ * (CAP + 1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for nres in (10, 5):
        L.check(lib.mk_qset_run_compact(ix._h, qs, nres, 10, 100.0 if nres == 10 else 200.0, CAP, rows.data_ptr()))
        L.check(lib.mk_sync(ix._h))
        n = (rows.view(NQ, CAP + 1)[:, 0].cpu().numpy() & 0xFFFFFFFF).astype(np.int64)
        cap = entrant_cap(nres, G)
        print(f"G {G:6d} top-{nres:<2d}: entrants per query mean {n.mean():6.1f} sd {n.std():5.1f} p99 {np.percentile(n, 99):5.0f} "
              f"p99.9 {np.percentile(n, 99.9):5.0f} max {n.max():4d}; over 96 slots: {(n > 96).mean() * 100:5.2f} %; "
              f"entrant_cap = {cap}: over it {(n > cap).sum()} of {NQ}", flush=True)
    lib.mk_qset_free(ix._h, qs)
    ix.close()
