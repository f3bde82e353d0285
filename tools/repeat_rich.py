#!/usr/bin/env python3
"""Build rate on REPEAT-RICH genomes (every rate so far was measured on uniform-random sequence).
Tandem repeats and homopolymer runs make many k-mers of one stretch identical: they hash to ONE partition, i.e. many
lanes of the reduce kernel's per-partition atomic minimum meet on one LDS entry (profiles/r3_ubench.txt: 64 lanes on one
entry cost 620 cycles for ds_min_u32 against 5.6 for distinct ones), and of the scatter kernel's bin counters too.
Genomes: tests/synth.py tandem_rich (share of the length in repeats of a 1-60 base unit, 200-5,000 bases each), handed
over packed from page-locked buffers like the `miekki` binary does, into an index whose Bloom filter has filled up;
reported: device time per 64-genome batch by stage (HIP events around the kernels) and the wall rate.
    python tools/repeat_rich.py [genomes per share] [h]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import miekki_amd  # noqa: E402
from miekki_amd import lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
LEN = 5_000_000
lib = L.load_library()
ix = miekki_amd.Miekki(31, h, 8, 33, 200)
ix.reserve(3072 + 8 * N + 64)
ix.insert_synthetic(20_000_000, 3072, LEN)                     # a Bloom filter that has filled up
L.check(lib.mk_sync(ix._h))
cw, xw = lib.mk_pack_code_words(LEN), lib.mk_pack_except_words(LEN)
pk = C.c_void_p()
L.check(lib.mk_host_alloc(ix._h, 64 * (cw + xw) * 8, C.byref(pk)))
for share in (0.0, 0.05, 0.1, 0.2, 0.5):
    rates = []
    ix.reset_stats()
    dev_t0 = ix.stats()
    wall = 0.0
    for b0 in range(0, N, 64):
        n = min(64, N - b0)
        arr = (L.PackedSeq * n)()
        for i in range(n):                                       # packed outside the timed region (the binary's readers do it while parsing)
            seq = synth.tandem_rich(5000 + b0 + i, LEN, share) if share else synth.genome_bases(5000 + b0 + i, 0, LEN)
            codes = pk.value + i * (cw + xw) * 8
            exc = codes + cw * 8
            assert lib.mk_pack_append(codes, exc, 0, seq, LEN) == 0
            arr[i].codes, arr[i].except_, arr[i].len, arr[i].head = codes, None, LEN, seq[:32]
        t = time.perf_counter()
        L.check(lib.mk_index_append_packed(ix._h, arr, n))
        L.check(lib.mk_sync(ix._h))
        wall += time.perf_counter() - t
    st = ix.stats()
    nb = (N + 63) // 64
    print(f"repeat share {share:4.2f}: {N / wall:8.0f} sketches/s (one batch in flight at a time, PCIe included); per 64-genome batch: "
          f"scatter {st['build_sketch_ms'] / nb:6.2f} ms, reduce + rows + Bloom {st['build_finalize_ms'] / nb:6.2f} ms", flush=True)
lib.mk_host_free(ix._h, pk)
ix.close()
