#!/usr/bin/env python3
"""SURVEY 8f row N4, the question before any codec: how compressible ARE the fingerprint columns?
Builds G synthetic genomes on the GPU (the product path), exports a sample of matrix rows through
mk_index_export_columns and reports the byte entropy (order 0, and order 1 along a row = between
neighbouring genomes) next to what zlib -- the reference's codec, compress_index, Miekki.cpp:863-868 /
utils.cpp:321-360 -- and lzma make of the very same bytes.
    python tools/column_entropy.py [G] [h] [rows]"""
import lzma, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miekki_amd
from miekki_amd import lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
for fpb in (8, 16):
    ix = miekki_amd.Miekki(31, h, fpb, 33, 200)
    W = fpb // 8
    ix.reserve(G)
    ix.insert_synthetic(0, G, 5_000_000)
    P = 1 << h
    buf = np.empty(rows * G * W, np.uint8)
    p0 = P // 3 // rows * rows
    L.check(ix._lib.mk_index_export_columns(ix._h, p0, p0 + rows, buf.ctypes.data))
    ix.close()
    if W == 2:
        sym = buf.reshape(-1, 2).astype(np.uint32)
        sym = (sym[:, 0] << 8) | sym[:, 1]                         # big-endian pairs (Miekki.cpp:230-231)
    else:
        sym = buf.astype(np.uint32)
    hist = np.bincount(sym, minlength=1 << fpb).astype(np.float64)
    p = hist[hist > 0] / hist.sum()
    H0 = float(-(p * np.log2(p)).sum())
    # order 1 along a row: H(x_g | x_{g-1}) -- neighbouring genomes (independent here; related strains would not be)
    m = sym.reshape(rows, G)
    pair = (m[:, :-1].astype(np.int64) << fpb) | m[:, 1:]
    _, cnt = np.unique(pair, return_counts=True)
    pj = cnt / cnt.sum()
    H1 = float(-(pj * np.log2(pj)).sum()) - H0
    raw = buf.tobytes()
    print(f"{fpb}-bit fingerprints, {G} synthetic 5 Mb genomes, -h {h}, rows [{p0}, {p0 + rows}): {len(raw)} bytes")
    print(f"  order-0 entropy {H0:.3f} bits per fingerprint -> best static entropy coder {fpb / H0:.3f}:1"
          f"   (empty: {hist[-1] / hist.sum():.4f} of the values)")
    print(f"  order-1 entropy along a row {H1:.3f} bits (neighbouring genomes are independent in this collection)")
    # the reference's own codec: every column (= one row of the matrix here, G fingerprints) a zlib stream of its
    # own at level 1 (main.cpp:198 -> compress_index(1) -> compress_string, utils.cpp:321-360)
    per_row = sum(len(zlib.compress(raw[r * G * W:(r + 1) * G * W], 1)) for r in range(rows))
    print(f"  zlib level 1, one stream per column like compress_index(1): {len(raw) / per_row:.3f}:1")
    for name, fn in (("zlib level 1, one stream", lambda b: zlib.compress(b, 1)), ("zlib level 9, one stream", lambda b: zlib.compress(b, 9)),
                     ("lzma, one stream", lambda b: lzma.compress(b, preset=6))):
        n = len(fn(raw))
        print(f"  {name}: {len(raw) / n:.3f}:1")
