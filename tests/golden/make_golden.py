#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the authoring container: needs oracle/_ref/ (built by
`make -C oracle ref` from the unmodified sources under /root/reference).  Inputs
come from tests/synth.py; what is committed here are the reference's OUTPUTS
(scores, hit tuples, sizes, digests, out.txt text) -- data, never source.

    python tests/golden/make_golden.py [case ...]
"""
import gzip
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "_ref")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def read_hits(path, nq):
    """-> list of (genome u32[], matches u32[], jaccard f64[], inter f64[]) flattened with offsets"""
    data = open(path, "rb").read()
    pos, off = 0, [0]
    gen, mat, jac, inter = [], [], [], []
    for _ in range(nq):
        (n,) = struct.unpack_from("<I", data, pos); pos += 4
        for _ in range(n):
            g, m, j, i = struct.unpack_from("<IIdd", data, pos); pos += 24
            gen.append(g); mat.append(m); jac.append(j); inter.append(i)
        off.append(len(gen))
    assert pos == len(data)
    return (np.array(off, np.int64), np.array(gen, np.uint32), np.array(mat, np.uint32),
            np.array(jac, np.float64), np.array(inter, np.float64))


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode(errors="replace"))
        raise SystemExit(f"{cmd} failed with {r.returncode}")
    return r.stdout


def strip_timing(out: bytes) -> bytes:
    import re
    return re.sub(rb"elapsed time: [0-9.e+-]+s", b"elapsed time: Xs", out)


def parse_stream(raw: bytes, W: int):
    k, h, fpb, nbm, G, bl2 = struct.unpack_from("<6I", raw, 0)
    (bbits,) = struct.unpack_from("<Q", raw, 24)
    (thr,) = struct.unpack_from("<I", raw, 34)
    P = 1 << h
    pos = 39
    cols = np.frombuffer(raw, np.uint8, P * G * W, pos).reshape(P, G * W); pos += P * G * W
    gsz = np.frombuffer(raw, np.uint64, G, pos); pos += 8 * G
    bloom = np.frombuffer(raw, np.uint8, bbits // 8, pos); pos += bbits // 8
    ssz = np.frombuffer(raw, np.uint32, G, pos); pos += 4 * G
    assert pos == len(raw), (pos, len(raw))
    return dict(k=k, h=h, fpb=fpb, nbm=nbm, G=G, bl2=bl2, bbits=bbits, thr=thr, compressed=raw[38],
                cols=cols, gsz=gsz, bloom=bloom, ssz=ssz)


def make_case(name):
    case = synth.CASES[name]()
    sfx = "16" if case.fp_bits == 16 else ""
    harness = os.path.join(REFDIR, "ref_harness" + sfx)
    cli = os.path.join(REFDIR, "Miekki" + sfx)
    W = case.fp_bits // 8
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        for fn, data, gz in case.genome_files:
            with open(os.path.join(d, fn), "wb") as f:
                f.write(gzip.compress(data, 1) if gz else data)
        with open(os.path.join(d, "genomes.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
            f.write(b"missing_file.fa\nab\n")          # "Missed file" + a <=3 char line (555-557)
        with open(os.path.join(d, "queries.fa"), "wb") as f:
            for hd, sq in case.queries:
                f.write(hd + b"\n" + sq + b"\n")
        nsk = 6
        params = [str(case.k), str(case.h), str(case.f), str(case.b), str(case.threshold)]
        run([harness, "golden", ".", *params, str(nsk)], d)
        kept = case.query_sequences()
        nq = len(kept)
        raw = gzip.decompress(open(os.path.join(d, "harness_idx.gz"), "rb").read())
        st = parse_stream(raw, W)
        G, P = st["G"], 1 << case.h
        assert G == len(case.genome_sequences())
        scores = np.fromfile(os.path.join(d, "scores.u32"), np.uint32).reshape(nq, G)
        qscores = np.fromfile(os.path.join(d, "qseq_scores.u32"), np.uint32).reshape(nq, G)
        qactive = np.fromfile(os.path.join(d, "qseq_active.u32"), np.uint32)
        assert (scores == qscores).all(), "reference query_sequence != query_sequences"
        # per-query raw sketches: active + digest of fp(u16)[P] + hash(u64)[P]
        skq = open(os.path.join(d, "sketch_q.bin"), "rb").read()
        rec = 4 + 2 * P + 8 * P
        sk_active, sk_fp_sha, sk_hash_sha = [], [], []
        for i in range(min(nsk, nq)):
            r = skq[i * rec:(i + 1) * rec]
            sk_active.append(struct.unpack_from("<I", r, 0)[0])
            sk_fp_sha.append(sha(r[4:4 + 2 * P])); sk_hash_sha.append(sha(r[4 + 2 * P:]))
        out = dict(
            k=case.k, h=case.h, f=case.f, b=case.b, threshold=case.threshold, G=G, nq=nq,
            scores=scores, qseq_active=qactive,
            sketch_size=st["ssz"].copy(), genome_size=st["gsz"].copy(),
            col_sha_per_genome=np.array(
                [sha(np.ascontiguousarray(st["cols"][:, g * W:(g + 1) * W]).tobytes()) for g in range(G)]),
            cols_head=st["cols"][:64].copy(), cols_tail=st["cols"][-64:].copy(),
            bloom_nonzero=int(np.count_nonzero(st["bloom"])), bloom_sha=sha(st["bloom"].tobytes()),
            bloom_nonzero_idx_head=np.flatnonzero(st["bloom"])[:256].astype(np.uint64),
            bloom_nonzero_val_head=st["bloom"][np.flatnonzero(st["bloom"])[:256]].copy(),
            sk_active=np.array(sk_active, np.uint32), sk_fp_sha=np.array(sk_fp_sha),
            sk_hash_sha=np.array(sk_hash_sha),
        )
        masked = bytearray(raw); masked[32] = 0; masked[38] = 0
        out["stream_sha_masked"] = sha(bytes(masked))       # bytes 32 (uninitialised) and 38 (compressed flag) zeroed
        out["stream_len"] = len(raw)
        for tag in ("approx", "exact_a", "exact_A", "loose", "loose10"):
            off, g, m, j, i = read_hits(os.path.join(d, f"hits_{tag}.bin"), nq)
            out[f"hits_{tag}_off"], out[f"hits_{tag}_genome"], out[f"hits_{tag}_matches"] = off, g, m
            out[f"hits_{tag}_jaccard"], out[f"hits_{tag}_inter"] = j, i
        # real CLI, -t 1: approximate mode with dump, then exact mode
        base = ["-k", str(case.k), "-h", str(case.h), "-f", str(case.f), "-b", str(case.b),
                "-s", str(case.threshold), "-t", "1"]
        so = run([cli, "-l", "genomes.lst", "-a", "queries.fa", "-o", "out.txt", "-d", "idx.gz", *base], d)
        out_txt = open(os.path.join(d, "out.txt"), "rb").read()
        ref_idx_gz = open(os.path.join(d, "idx.gz"), "rb").read()
        raw2 = bytearray(gzip.decompress(ref_idx_gz))
        raw2[32] = 0; raw2[38] = 0
        assert sha(bytes(raw2)) == out["stream_sha_masked"], "CLI dump != harness dump"
        so_i = run([cli, "-i", "idx.gz", "-a", "queries.fa", "-o", "out_i.txt", "-t", "1"], d)
        assert open(os.path.join(d, "out_i.txt"), "rb").read() == out_txt, "-i path differs from -l path"
        so_e = run([cli, "-l", "genomes.lst", "-a", "queries.fa", "-e", "-o", "exact.txt", *base], d)
        exact_txt = open(os.path.join(d, "exact.txt"), "rb").read()
        # whole-file queries (-A): each genome file as one query
        with open(os.path.join(d, "qfiles.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
        run([cli, "-i", "idx.gz", "-A", "qfiles.lst", "-o", "outA.txt", "-t", "1"], d)
        outA_txt = open(os.path.join(d, "outA.txt"), "rb").read()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    if name in synth.REF_INDEX_CASES:                    # the reference's own -d output, byte for byte (a data fixture)
        with open(os.path.join(HERE, f"{name}_ref_idx.gz"), "wb") as f:
            f.write(ref_idx_gz)
    for fn, data in ((f"{name}_out.txt", out_txt), (f"{name}_exact.txt", exact_txt),
                     (f"{name}_outA.txt", outA_txt),
                     (f"{name}_stdout_l.txt", strip_timing(so)),
                     (f"{name}_stdout_i.txt", strip_timing(so_i)),
                     (f"{name}_stdout_e.txt", strip_timing(so_e))):
        with open(os.path.join(HERE, fn), "wb") as f:
            f.write(data)
    print(f"{name}: G={G} nq={nq} stream={len(raw)}B out.txt={len(out_txt)}B exact={len(exact_txt)}B")


def make_exact_whole_file(name):
    """`-l … -A … -e` (query_file_of_file_exact, Miekki.cpp:616-645, 763-788): every
    genome file of the case as one whole-file query in exact mode."""
    case = synth.CASES[name]()
    sfx = "16" if case.fp_bits == 16 else ""
    cli = os.path.join(REFDIR, "Miekki" + sfx)
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        for fn, data, gz in case.genome_files:
            with open(os.path.join(d, fn), "wb") as f:
                f.write(gzip.compress(data, 1) if gz else data)
        with open(os.path.join(d, "genomes.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
        base = ["-k", str(case.k), "-h", str(case.h), "-f", str(case.f), "-b", str(case.b),
                "-s", str(case.threshold), "-t", "1"]
        run([cli, "-l", "genomes.lst", "-A", "genomes.lst", "-e", "-o", "exactA.txt", *base], d)
        txt = open(os.path.join(d, "exactA.txt"), "rb").read()
    with open(os.path.join(HERE, f"{name}_exactA.txt"), "wb") as f:
        f.write(txt)
    print(f"{name}: -A -e output {len(txt)}B, {len(txt.splitlines())} lines")


def make_exact_only(name):
    """`-l … -a … -e` of an EXTRA_CASES entry: only the exact-mode output file is kept."""
    case = synth.EXTRA_CASES[name]()
    cli = os.path.join(REFDIR, "Miekki" + ("16" if case.fp_bits == 16 else ""))
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        for fn, data, gz in case.genome_files:
            with open(os.path.join(d, fn), "wb") as f:
                f.write(gzip.compress(data, 1) if gz else data)
        with open(os.path.join(d, "genomes.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
        with open(os.path.join(d, "queries.fa"), "wb") as f:
            f.write(b"".join(h + b"\n" + s + b"\n" for h, s in case.queries))
        base = ["-k", str(case.k), "-h", str(case.h), "-f", str(case.f), "-b", str(case.b),
                "-s", str(case.threshold), "-t", "1"]
        run([cli, "-l", "genomes.lst", "-a", "queries.fa", "-e", "-o", "exact.txt", *base], d)
        txt = open(os.path.join(d, "exact.txt"), "rb").read()
    with open(os.path.join(HERE, f"{name}_exact.txt"), "wb") as f:
        f.write(txt)
    print(f"{name}: -a -e output {len(txt)}B, {len(txt.splitlines())} lines")


def make_out_only(name):
    """`-l … -a … -o out.txt` (and `-A`) of an EXTRA_CASES entry: only the output files are kept."""
    case = synth.EXTRA_CASES[name]()
    cli = os.path.join(REFDIR, "Miekki" + ("16" if case.fp_bits == 16 else ""))
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        seen = set()
        for fn, data, gz in case.genome_files:
            if fn not in seen:
                seen.add(fn)
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(gzip.compress(data, 1) if gz else data)
        with open(os.path.join(d, "genomes.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
        with open(os.path.join(d, "queries.fa"), "wb") as f:
            f.write(b"".join(h + b"\n" + s + b"\n" for h, s in case.queries))
        base = ["-k", str(case.k), "-h", str(case.h), "-f", str(case.f), "-b", str(case.b),
                "-s", str(case.threshold), "-t", "1"]
        run([cli, "-l", "genomes.lst", "-a", "queries.fa", "-o", "out.txt", *base], d)
        txt = open(os.path.join(d, "out.txt"), "rb").read()
    with open(os.path.join(HERE, f"{name}_out.txt"), "wb") as f:
        f.write(txt)
    print(f"{name}: -a output {len(txt)}B, {len(txt.splitlines())} lines")


def make_single(name):
    """The case's genome files through the reference's index_file / insert_sequence (Miekki.cpp:518-536, 243-273) one by
    one: the sizes and the digest of the index stream that one-genome path leaves."""
    case = synth.CASES[name]()
    sfx = "16" if case.fp_bits == 16 else ""
    harness = os.path.join(REFDIR, "ref_harness" + sfx)
    W = case.fp_bits // 8
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        for fn, data, gz in case.genome_files:
            with open(os.path.join(d, fn), "wb") as f:
                f.write(gzip.compress(data, 1) if gz else data)
        with open(os.path.join(d, "genomes.lst"), "wb") as f:
            f.write(b"".join(fn.encode() + b"\n" for fn, _, _ in case.genome_files))
            f.write(b"missing_file.fa\n")
        run([harness, "single", d, str(case.k), str(case.h), str(case.f), str(case.b), str(case.threshold)], d)
        raw = gzip.decompress(open(os.path.join(d, "single_idx.gz"), "rb").read())
    st = parse_stream(raw, W)
    assert st["G"] == len(case.genome_sequences())
    masked = bytearray(raw); masked[32] = 0; masked[38] = 0
    batch = np.load(os.path.join(HERE, f"{name}.npz"))
    np.savez_compressed(os.path.join(HERE, f"{name}_single.npz"), G=st["G"], genome_size=st["gsz"].copy(),
                        sketch_size=st["ssz"].copy(), stream_sha_masked=sha(bytes(masked)), stream_len=len(raw))
    differ = int((batch["genome_size"] != st["gsz"]).sum())
    print(f"{name}_single: G={st['G']} genome_size differs from insert_sequences' for {differ} genomes")


def make_filter_cases():
    """Synthetic filter_results inputs built to hit heap ties and replacement."""
    rng = np.random.default_rng(20261003)
    cases = []
    for c in range(60):
        G = int(rng.integers(1, 80))
        nres = int(rng.integers(1, 12))
        ms = int(rng.integers(0, 4))
        ss = rng.integers(1, 5, G).astype(np.uint32) * 100        # few distinct values -> many ties
        gs = rng.integers(1, 4, G).astype(np.uint64) * 1000
        sc = rng.integers(0, 6, G).astype(np.uint32)
        mi = float(rng.choice([0.0, 5.0, 10.0, 20.0]))
        cases.append((G, nres, ms, mi, ss, gs, sc))
    with tempfile.TemporaryDirectory(prefix="mkgold_") as d:
        with open(os.path.join(d, "in.bin"), "wb") as f:
            f.write(struct.pack("<I", len(cases)))
            for G, nres, ms, mi, ss, gs, sc in cases:
                f.write(struct.pack("<IIId", G, nres, ms, mi))
                f.write(ss.tobytes()); f.write(gs.tobytes()); f.write(sc.tobytes())
        run([os.path.join(REFDIR, "ref_harness"), "filter", "in.bin", "out.bin"], d)
        off, g, m, j, i = read_hits(os.path.join(d, "out.bin"), len(cases))
    out = dict(n=len(cases), off=off, genome=g, matches=m, jaccard=j, inter=i)
    for c, (G, nres, ms, mi, ss, gs, sc) in enumerate(cases):
        out[f"c{c}_par"] = np.array([G, nres, ms], np.int64); out[f"c{c}_mi"] = np.float64(mi)
        out[f"c{c}_ss"], out[f"c{c}_gs"], out[f"c{c}_sc"] = ss, gs, sc
    np.savez_compressed(os.path.join(HERE, "filter_ties.npz"), **out)
    print(f"filter_ties: {len(cases)} cases, {len(g)} hits")


if __name__ == "__main__":
    names = sys.argv[1:] or (list(synth.CASES) + ["filter", "exactA:messy", "exactA:h20", "exactA:w16", "exact:flush", "out:dups",
                                 "single:h16z", "single:h20", "single:w16", "single:messy"])
    for n in names:
        if n == "filter":
            make_filter_cases()
        elif n.startswith("single:"):
            make_single(n.split(":", 1)[1])
        elif n.startswith("out:"):
            make_out_only(n.split(":", 1)[1])
        elif n.startswith("exact:"):
            make_exact_only(n.split(":", 1)[1])
        elif n.startswith("exactA:"):
            make_exact_whole_file(n.split(":", 1)[1])
        else:
            make_case(n)
