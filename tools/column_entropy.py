#!/usr/bin/env python3
"""SURVEY 8f row N4, the question before any codec: how compressible ARE the fingerprint columns?
Builds a collection on the GPU (the product path), exports a sample of matrix rows through
mk_index_export_columns and reports the byte entropy (order 0, and order 1 along a row = between
neighbouring genomes) next to what zlib -- the reference's codec, compress_index, Miekki.cpp:863-868 /
utils.cpp:321-360 -- and lzma make of the very same bytes, and what the codec a GPU could decode at line rate
makes of them: "same as the genome before" bits + the differing fingerprints (delta vs previous + bit packing).

    python tools/column_entropy.py [G] [h] [rows]                      independent synthetic genomes
    python tools/column_entropy.py strains SPECIES STRAINS RATE [h] [rows] [length]
        SPECIES x STRAINS related genomes (tests/synth.py: strain; RATE substitutions per base vs the species genome),
        rows in LIST order species by species (a similarity order: what a user who sorts the -l list gets) and in a
        SHUFFLED order (what an arbitrary list gives) -- README.md:136-138: "A clever ordering of the lines could allow
        a very efficient column compression"."""
import lzma, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import miekki_amd
from miekki_amd import lib as L


def report(title, buf, rows, G, fpb):
    W = fpb // 8
    if W == 2:
        sym = buf.reshape(-1, 2).astype(np.uint32)
        sym = (sym[:, 0] << 8) | sym[:, 1]                         # big-endian pairs (Miekki.cpp:230-231)
    else:
        sym = buf.astype(np.uint32)
    hist = np.bincount(sym, minlength=1 << fpb).astype(np.float64)
    p = hist[hist > 0] / hist.sum()
    H0 = float(-(p * np.log2(p)).sum())
    m = sym.reshape(rows, G)
    pair = (m[:, :-1].astype(np.int64) << fpb) | m[:, 1:]
    _, cnt = np.unique(pair, return_counts=True)
    pj = cnt / cnt.sum()
    H1 = float(-(pj * np.log2(pj)).sum()) - H0
    raw = buf.tobytes()
    print(f"{title}: {len(raw)} bytes")
    print(f"  order-0 entropy {H0:.3f} bits per fingerprint -> best static entropy coder {fpb / H0:.3f}:1"
          f"   (empty: {hist[-1] / hist.sum():.4f} of the values)")
    print(f"  order-1 entropy along a row (given the genome before) {H1:.3f} bits -> {fpb / max(H1, 1e-9):.3f}:1")
    # delta vs the previous genome + bit packing, in pieces of 1024 genomes that decode on their own (a scan tile):
    # one "differs from the genome before" bit per fingerprint (the first of a piece always differs) + the differing ones raw
    same = np.zeros((rows, G), bool)
    same[:, 1:] = m[:, 1:] == m[:, :-1]
    same[:, ::1024] = False
    literals = int((~same).sum())
    packed = rows * G / 8 + literals * W
    print(f"  'same as the genome before' bit + differing fingerprints raw ({(~same).mean() * 100:.1f} % differ): {len(raw) / packed:.3f}:1")
    per_row = sum(len(zlib.compress(raw[r * G * W:(r + 1) * G * W], 1)) for r in range(rows))
    print(f"  zlib level 1, one stream per column like compress_index(1): {len(raw) / per_row:.3f}:1")
    for name, fn in (("zlib level 9, one stream", lambda b: zlib.compress(b, 9)), ("lzma, one stream", lambda b: lzma.compress(b, preset=6))):
        print(f"  {name}: {len(raw) / len(fn(raw)):.3f}:1")
    return len(raw) / packed


def export_rows(ix, p0, rows, G, W):
    buf = np.empty(rows * G * W, np.uint8)
    L.check(ix._lib.mk_index_export_columns(ix._h, p0, p0 + rows, buf.ctypes.data))
    return buf


if len(sys.argv) > 1 and sys.argv[1] == "strains":
    import synth
    ns, nt, rate = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    h = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    rows = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
    length = int(sys.argv[7]) if len(sys.argv) > 7 else 5_000_000
    G = ns * nt
    order = [(s, t) for s in range(ns) for t in range(nt)]
    shuffled = list(order)
    np.random.default_rng(1).shuffle(shuffled)
    genomes = {st: synth.strain(1000 + st[0], st[1], length, rate) for st in order}      # generated once (host, numpy)
    for fpb in (8, 16):
        for name, lst in (("list order = species by species", order), ("shuffled list", shuffled)):
            ix = miekki_amd.Miekki(31, h, fpb, 33, 200)
            ix.reserve(G)
            for i in range(0, G, 32):
                ix.insert_sequences([genomes[st] for st in lst[i:i + 32]])
            P = 1 << h
            p0 = P // 3 // rows * rows
            buf = export_rows(ix, p0, rows, G, fpb // 8)
            ix.close()
            report(f"{fpb}-bit fingerprints, {ns} species x {nt} strains at {rate * 100:g} % substitutions ({length} bases), {name}, -h {h}, "
                   f"rows [{p0}, {p0 + rows})", buf, rows, G, fpb)
    sys.exit(0)

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
for fpb in (8, 16):
    ix = miekki_amd.Miekki(31, h, fpb, 33, 200)
    ix.reserve(G)
    ix.insert_synthetic(0, G, 5_000_000)
    P = 1 << h
    p0 = P // 3 // rows * rows
    buf = export_rows(ix, p0, rows, G, fpb // 8)
    ix.close()
    report(f"{fpb}-bit fingerprints, {G} independent synthetic 5 Mb genomes, -h {h}, rows [{p0}, {p0 + rows})", buf, rows, G, fpb)
