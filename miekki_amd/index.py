"""Python mirror of the reference's `Miekki` class for the hot path (Miekki.h:34-137).

Same member names, argument meaning and error behaviour as the reference so that
parity tests read like calls into it.  Every computation goes through the C ABI
of libmiekki_hip.so; when the library or the GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import struct
from collections import namedtuple

import numpy as np

from . import lib as L

# Miekki.h:27-31
SimilarityScore = namedtuple("SimilarityScore", "genome matches jaccard intersection")

_HDR = struct.Struct("<6IQBBIB")     # 39 bytes, unpadded
assert _HDR.size == 39


class Miekki:
    def __init__(self, kmer_size=31, number_minimizer_log2=17, number_bit_minimizer=8,
                 bloom_size_log2=33, threshold=200, device=0, genome_id_base=0):
        self._lib = L.load_library()
        self._p = L.Params(kmer_size, number_minimizer_log2, number_bit_minimizer, bloom_size_log2,
                           int(threshold), device, genome_id_base, 0)
        h = C.c_void_p()
        st = self._lib.mk_create(C.byref(self._p), C.byref(h))
        if st == -2:
            raise NotImplementedError("not implemented")          # Miekki.cpp:235-237
        L.check(st)
        self._h = h
        self.file_names = []                                       # Miekki.h:59 (never persisted)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mk_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- members of the reference object
    kmer_size = property(lambda s: s._p.k)
    number_minimizer_log2 = property(lambda s: s._p.h)
    number_minimizer = property(lambda s: 1 << s._p.h)
    number_bit_minimizer = property(lambda s: s._p.fp_bits)
    bloom_size_log2 = property(lambda s: s._p.bloom_log2)
    bloom_size = property(lambda s: (1 << s._p.bloom_log2) if s._p.bloom_log2 else 0)
    threshold = property(lambda s: s._p.threshold)
    index_size = property(lambda s: s._lib.mk_index_size(s._h))
    W = property(lambda s: s._p.fp_bits // 8)

    @property
    def genome_size(self):
        out = np.zeros(self.index_size, np.uint64)
        L.check(self._lib.mk_index_export_sizes(self._h, out.ctypes.data, None))
        return out

    @property
    def sketch_size(self):
        out = np.zeros(self.index_size, np.uint32)
        L.check(self._lib.mk_index_export_sizes(self._h, None, out.ctypes.data))
        return out

    def reserve(self, n_genomes):
        L.check(self._lib.mk_reserve(self._h, n_genomes))

    def compress_index(self):
        """Miekki::compress_index (Miekki.cpp:863-868) for the rows beyond the matrix's HBM budget: returns
        (raw bytes, packed bytes) of those rows -- equal when nothing was packed (no cold rows, or nothing to gain)."""
        raw, packed = C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.mk_index_compress(self._h, C.byref(raw), C.byref(packed)))
        return raw.value, packed.value

    def decompress_index(self):
        """Miekki::decompress_index (Miekki.cpp:872-877)."""
        L.check(self._lib.mk_index_decompress(self._h))

    def insert_synthetic_strains(self, first_id, n, length, strains, rate_ppm):
        L.check(self._lib.mk_index_append_synthetic_strains(self._h, first_id, n, length, strains, rate_ppm))
        self.file_names += [f"strain:{first_id + i}" for i in range(n)]

    def stats(self):
        s = L.Stats()
        L.check(self._lib.mk_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.Stats._fields_}

    def reset_stats(self):
        L.check(self._lib.mk_reset_stats(self._h))

    # ---- index build (Miekki.cpp:277-314, 540-588)
    def insert_sequence(self, seq, title=""):
        """Miekki::insert_sequence (Miekki.cpp:243-273): one genome, its own size estimate (no u32 wrap of active^2)."""
        seq = bytes(seq)
        L.check(self._lib.mk_index_insert_sequence(self._h, seq, len(seq)))
        self.file_names.append(title)

    def index_file(self, path):
        """Miekki::index_file (Miekki.cpp:518-536): every non-header line of a (gzip'd) FASTA file as ONE genome."""
        import gzip
        import os
        if not os.path.exists(path):
            print(f"Missed file: {path}")
            return
        raw = open(path, "rb").read()
        if raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)
        ref = b"".join(ln for ln in raw.split(b"\n") if not ln.startswith(b">"))
        if len(ref) >= self.kmer_size:
            self.insert_sequence(ref, path)

    def insert_sequences(self, seqs, names=None):
        seqs = [bytes(s) for s in seqs]
        if not seqs:
            return
        ptrs, lens = L.seq_arrays(seqs)
        L.check(self._lib.mk_index_append(self._h, ptrs, lens, len(seqs)))
        self.file_names += list(names) if names else [""] * len(seqs)

    def insert_sequences_packed(self, seqs, names=None):
        """insert_sequences for sequences handed over in the packed form of mk_index_append_packed
        (2-bit codes + exception bits + the first 32 characters): same index, a quarter of the bytes."""
        packed = [pack_sequence(bytes(s)) for s in seqs]
        if not packed:
            return
        arr = (L.PackedSeq * len(packed))()
        for i, (codes, exc, n, head) in enumerate(packed):
            arr[i].codes = codes.ctypes.data
            arr[i].except_ = exc.ctypes.data if exc is not None else None
            arr[i].len = n
            arr[i].head = head
        L.check(self._lib.mk_index_append_packed(self._h, arr, len(packed)))
        self.file_names += list(names) if names else [""] * len(packed)

    def insert_gz_files(self, blobs, names=None, fallback=True):
        """index_file_of_file's loop (Miekki.cpp:559-573) over gzip'd genome files whose BYTES are handed over: inflated,
        stripped of header lines and line feeds and appended on the device (mk_gz_unpack; the sequences never visit the
        host); a file the device refuses is inflated here (zstr's semantics: every member) when `fallback` is set.
        Returns the files' statuses (mk_gz_status).  Sequences shorter than k are skipped like the reference's."""
        blobs = [bytes(b) for b in blobs]
        n = len(blobs)
        if not n:
            return []
        ptrs, lens = L.seq_arrays(blobs)
        batch = C.c_void_p()
        L.check(self._lib.mk_gz_unpack(self._h, ptrs, lens, n, C.byref(batch)))
        status = []
        try:
            run, kept_names = [], []          # run: files of the batch waiting to be appended together
            def flush_run():
                for g0 in range(0, len(run), 64):
                    part = run[g0:g0 + 64]
                    L.check(self._lib.mk_index_append_gz(self._h, batch, (C.c_uint32 * len(part))(*part), len(part)))
                del run[:]
            for i in range(n):
                ln, st = C.c_uint64(), C.c_int32()
                L.check(self._lib.mk_gz_sequence(batch, i, C.byref(ln), C.byref(st)))
                status.append(st.value)
                if st.value == 0:
                    if ln.value >= self.kmer_size:
                        run.append(i); kept_names.append(names[i] if names else "")
                elif fallback:
                    ref = b"".join(l for l in _gunzip_members(blobs[i]).split(b"\n") if not l.startswith(b">"))
                    if len(ref) >= self.kmer_size:
                        flush_run()                       # (list order: the files before it first)
                        L.check(self._lib.mk_index_append(self._h, (C.c_char_p * 1)(ref), (C.c_uint64 * 1)(len(ref)), 1))
                        kept_names.append(names[i] if names else "")
            flush_run()
            L.check(self._lib.mk_sync(self._h))
            self.file_names += kept_names
        finally:
            self._lib.mk_gz_free(batch)
        return status

    def insert_synthetic(self, first_id, n, length):
        L.check(self._lib.mk_index_append_synthetic(self._h, first_id, n, length))
        self.file_names += [f"synthetic:{first_id + i}" for i in range(n)]

    def index_file_of_file(self, path, log=print):
        """Miekki.cpp:540-588 at -t 1: one path per line, <=3 character lines ignored,
        missing files reported, sequences shorter than k skipped, flush every 11."""
        if not os.path.exists(path):
            log(f"Missed file of file: {path}")
            return
        batch, names = [], []
        for fn in _read_text(path).split(b"\n"):
            if len(fn) <= 3:
                continue
            fn = fn.decode()
            if not os.path.exists(fn):
                log(f"Missed file: {fn}")
                continue
            ref = b"".join(l for l in _read_text(fn).split(b"\n") if not l.startswith(b">"))
            if len(ref) >= self.kmer_size:
                batch.append(ref); names.append(fn)
                if len(batch) > 10:
                    self.insert_sequences(batch, names)
                    batch, names = [], []
        self.insert_sequences(batch, names)
        log(f"Reference indexed: {self.index_size}")

    # ---- queries
    def query_sequences(self, seqs):
        """Miekki.cpp:344-372 -> uint32 [len(seqs), index_size]."""
        seqs = [bytes(s) for s in seqs]
        out = np.zeros((len(seqs), self.index_size), np.uint32)
        if seqs and self.index_size:
            ptrs, lens = L.seq_arrays(seqs)
            L.check(self._lib.mk_query_scores(self._h, ptrs, lens, len(seqs), out.ctypes.data))
        return out

    def query_sequence(self, seq):
        """Miekki.cpp:318-340 -> (scores[index_size], active_minimizer)."""
        hits, active = self.query([seq], 0, 0, 0.0)
        return self.query_sequences([seq])[0], int(active[0])

    def query(self, seqs, nresults=10, min_score=10, min_intersection=None):
        """filter_results(query_sequences(batch), ...) in one device pass
        (Miekki.cpp:437).  Returns (list of hit lists, active partitions)."""
        if min_intersection is None:
            min_intersection = 0.5 * self.threshold
        seqs = [bytes(s) for s in seqs]
        nq = len(seqs)
        hits = (L.Hit * max(nq * max(nresults, 1), 1))()
        nhits = np.zeros(nq, np.uint32)
        active = np.zeros(nq, np.uint32)
        if nq:
            ptrs, lens = L.seq_arrays(seqs)
            L.check(self._lib.mk_query(self._h, ptrs, lens, nq, nresults, min_score, float(min_intersection),
                                       hits, nhits.ctypes.data, active.ctypes.data))
        out = []
        for q in range(nq):
            row = [hits[q * nresults + i] for i in range(int(nhits[q]))]
            out.append([SimilarityScore(h.genome, h.matches, h.jaccard, h.intersection) for h in row])
        return out, active

    def filter_results(self, scores, nresults, min_score, min_intersection):
        """Miekki.cpp:376-422 over host scores (a row or a matrix)."""
        scores = np.asarray(scores, np.uint32)
        if scores.ndim == 2:
            return [self.filter_results(r, nresults, min_score, min_intersection) for r in scores]
        ss, gs = self.sketch_size, self.genome_size
        with np.errstate(divide="ignore", invalid="ignore"):
            jac = scores.astype(np.float64) / ss
            inter = jac * gs
        keep = np.flatnonzero((scores >= min_score) & ~(inter < min_intersection))
        cand = (L.Hit * max(len(keep), 1))()
        for i, g in enumerate(keep):
            cand[i] = L.Hit(int(g) + self._p.genome_id_base, int(scores[g]), float(jac[g]), float(inter[g]))
        out = (L.Hit * max(nresults, 1))()
        n = self._lib.mk_filter_candidates(cand, len(keep), nresults, out)
        return [SimilarityScore(out[i].genome, out[i].matches, out[i].jaccard, out[i].intersection) for i in range(n)]

    @staticmethod
    def format_hits(name: bytes, hits) -> bytes:
        """One output line of query_file (Miekki.cpp:440-444)."""
        return name + b":" + b"".join(
            b"%d\t%d\t%d\t%s;" % (h.genome, h.matches, int(h.intersection), ("%f" % h.jaccard).encode())
            for h in hits) + b"\n"

    def query_file(self, path, out, batch_size=4096):
        """Miekki.cpp:426-483 at -t 1: strict 2-line records, records shorter than k
        skipped, one line per kept record."""
        if not os.path.exists(path):
            print("File problem")
            return
        lines = _read_text(path).split(b"\n")
        recs = [(lines[i], lines[i + 1] if i + 1 < len(lines) else b"") for i in range(0, len(lines), 2)]
        recs = [(h, s) for h, s in recs if len(s) >= self.kmer_size]
        for i in range(0, len(recs), batch_size):
            chunk = recs[i:i + batch_size]
            hits, _ = self.query([s for _, s in chunk], 10, 10, 0.5 * self.threshold)
            out.write(b"".join(self.format_hits(h, r) for (h, _), r in zip(chunk, hits)))

    # ---- exact mode (Miekki.cpp:792-859)
    def ground_truth_batch(self, queries, fasta: bytes):
        """|A n B| and |A u B| of each query against the genome file's k-mer set."""
        k = self.kmer_size
        contigs, ref = [], b""
        for line in fasta.split(b"\n"):
            if line.startswith(b">"):
                if len(ref) >= k:                       # short contigs leak into the next (806-812)
                    contigs.append(ref); ref = b""
            else:
                ref += line
        if len(ref) >= k:
            contigs.append(ref)
        queries = [bytes(q) for q in queries]
        cp, cl = L.seq_arrays(contigs)
        qp, ql = L.seq_arrays(queries)
        inter = np.zeros(len(queries), np.uint64)
        uni = np.zeros(len(queries), np.uint64)
        L.check(self._lib.mk_exact(self._h, cp, cl, len(contigs), qp, ql, len(queries),
                                   inter.ctypes.data, uni.ctypes.data))
        return inter, uni

    # ---- persistence (Miekki.cpp:649-719)
    def serialize(self, chunk_rows=4096):
        """Yield the uncompressed index stream piecewise."""
        G, W, P = self.index_size, self.W, self.number_minimizer
        yield _HDR.pack(self._p.k, self._p.h, self._p.fp_bits, 5, G, self._p.bloom_log2, self.bloom_size,
                        0, 0, self._p.threshold, 1)
        rows = max(1, min(P, (64 << 20) // max(G * W, 1)))
        buf = np.empty(rows * G * W, np.uint8)
        for p in range(0, P, rows):
            r = min(rows, P - p)
            if G:
                L.check(self._lib.mk_index_export_columns(self._h, p, p + r, buf.ctypes.data))
            yield buf[:r * G * W].tobytes()
        yield self.genome_size.tobytes()
        nb = self.bloom_size // 8
        step = 64 << 20
        b = np.empty(min(step, max(nb, 1)), np.uint8)
        for o in range(0, nb, step):
            e = min(nb, o + step)
            L.check(self._lib.mk_index_export_bloom(self._h, o, e, b.ctypes.data))
            yield b[:e - o].tobytes()
        yield self.sketch_size.tobytes()

    def dump_disk(self, path):
        """gzip level 1 like zstr::ofstream (zstr.hpp:82,230)."""
        with gzip.open(path, "wb", compresslevel=1) as f:
            for piece in self.serialize():
                f.write(piece)

    @classmethod
    def load(cls, path, device=0, genome_id_base=0):
        """Miekki(const string& input), Miekki.cpp:682-719; gzip or plain like zstr::ifstream."""
        if not os.path.exists(path):
            raise FileNotFoundError("File problem")
        with open(path, "rb") as raw:
            magic = raw.read(2)
        f = gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")
        with f:
            k, h, fpb, _nbm, G, bl2, bbits, _jac, _cont, thr, _comp = _HDR.unpack(_readn(f, 39))
            ix = cls(k, h, fpb, bl2, thr, device, genome_id_base)
            W, P = fpb // 8, 1 << h
            L.check(ix._lib.mk_index_import_begin(ix._h, G))
            rows = max(1, min(P, (64 << 20) // max(G * W, 1)))
            for p in range(0, P, rows):
                r = min(rows, P - p)
                buf = _readn(f, r * G * W)
                if G:
                    L.check(ix._lib.mk_index_import_columns(ix._h, p, p + r, buf))
            gs = np.frombuffer(_readn(f, 8 * G), np.uint64)
            nb, step = bbits // 8, 64 << 20
            for o in range(0, nb, step):
                e = min(nb, o + step)
                L.check(ix._lib.mk_index_import_bloom(ix._h, o, e, _readn(f, e - o)))
            ss = np.frombuffer(_readn(f, 4 * G), np.uint32)
            L.check(ix._lib.mk_index_import_sizes(ix._h, gs.ctypes.data if G else None, ss.ctypes.data if G else None)
                    if G else 0)
        return ix


def pack_sequence(seq: bytes, piece=None):
    """(codes uint64[], except uint64[] or None, len, head) of mk_packed_seq through the library's own
    host packer, mk_pack_append -- in `piece`-sized appends when given (a reader packs line by line)."""
    lib = L.load_library()
    n = len(seq)
    codes = np.zeros(lib.mk_pack_code_words(n), np.uint64)
    exc = np.zeros(lib.mk_pack_except_words(n), np.uint64)
    dirty = 0
    step = piece or max(n, 1)
    for at in range(0, n, step):
        chunk = seq[at:at + step]
        rc = lib.mk_pack_append(codes.ctypes.data, exc.ctypes.data, at, chunk, len(chunk))
        if rc < 0:
            raise ValueError("mk_pack_append failed")
        dirty |= rc
    return codes, (exc if dirty else None), n, seq[:32]


def _readn(f, n):
    b = f.read(n)
    if len(b) != n:
        raise EOFError("truncated index stream")
    return b


def _gunzip_members(data: bytes) -> bytes:
    """every gzip member from the start; what follows the last whole member is ignored (zlib's gzread does the same)"""
    import zlib
    out = []
    while data[:2] == b"\x1f\x8b":
        d = zlib.decompressobj(31)
        try:
            out.append(d.decompress(data))
        except zlib.error:
            break
        if not d.eof:
            break
        data = d.unused_data
    return b"".join(out)


def _read_text(path) -> bytes:
    """Whole file, gunzipped when it starts with the gzip magic (zstr.hpp:157-167)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)          # every member, like zstr (zstr.hpp:186-190)
    return data
