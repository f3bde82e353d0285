#!/usr/bin/env python3
"""What the column codec buys a collection that does not fit the GPU (mk_index_compress, cold.hip):
a collection of RELATED genomes (the device's strain generator: species by species in the list) at -h 20 whose matrix
exceeds its HBM budget, queried with the rows as they are and then packed.  A step is the bench's: sketch + gate + scan +
selection of Q synthetic 1 kb queries; the cold partition ranges cross PCIe once per query chunk -- raw, or packed.
    MIEKKI_HBM_MATRIX_MIB=<budget> python tools/bench_cold_codec.py [genomes] [strains per species] [rate ppm] [queries]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import miekki_amd  # noqa: E402
from miekki_amd import lib as L  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
RATE = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 20_000
LEN = 5_000_000
lib = L.load_library()
ix = miekki_amd.Miekki(31, 20, 8, 33, 200)
ix.reserve(G)
t = time.time()
for g0 in range(0, G, 4096):
    ix.insert_synthetic_strains(g0, min(4096, G - g0), LEN, S, RATE)
L.check(lib.mk_sync(ix._h))
build_s = time.time() - t
qs = C.c_void_p()
# queries cut from the species' own genomes (= strain 0 of every species: synthetic genome `species`)
L.check(lib.mk_qset_synthetic(ix._h, 0, Q, max(G // S, 1), LEN, 1000, C.byref(qs)))
cap = 128
rows = torch.zeros(Q * (cap + 1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()


def step():
    L.check(lib.mk_qset_invalidate(ix._h, qs))
    L.check(lib.mk_qset_run_compact(ix._h, qs, 10, 10, 100.0, cap, rows.data_ptr()))
    L.check(lib.mk_sync(ix._h))


def timed(n=3):
    step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return (time.perf_counter() - t0) / n


raw_step = timed()
ref = rows.clone()
t = time.time()
raw, packed = ix.compress_index()
pack_s = time.time() - t
packed_step = timed()
same = bool(torch.equal(ref, rows))
t = time.time()
ix.decompress_index()
unpack_s = time.time() - t
print(json.dumps({"workload": f"{G} genomes = {G // S} species x {S} strains at {RATE} ppm substitutions, -h 20, 8-bit fingerprints, "
                              f"{Q} x 1 kb queries, HBM budget {os.environ.get('MIEKKI_HBM_MATRIX_MIB', 'none')} MiB",
                  "build_s": build_s, "cold_rows_bytes": raw, "packed_bytes": packed, "ratio": raw / max(packed, 1),
                  "pack_s": pack_s, "unpack_s": unpack_s, "s_per_step_rows_as_they_are": raw_step, "s_per_step_packed": packed_step,
                  "step_shrinks_by": raw_step / packed_step, "same_exchange_rows": same}))
lib.mk_qset_free(ix._h, qs)
ix.close()
