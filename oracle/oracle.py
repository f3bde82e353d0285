"""ctypes binding of the CPU oracle (oracle/libmiekki_oracle.so).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  It is the checker, never the thing
measured or shipped.  The method names mirror the reference's Miekki class
(Miekki.h:99-135) so parity tests read like calls into the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmiekki_oracle.so")


class Hit(C.Structure):
    _fields_ = [("genome", C.c_uint32), ("matches", C.c_uint32),
                ("jaccard", C.c_double), ("intersection", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "lib"])


def _load():
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    vp, u32, u64, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p
    sig = {
        "mko_create": (vp, [u32, u32, u32, u32, u32]),
        "mko_destroy": (None, [vp]),
        "mko_index_size": (u32, [vp]), "mko_threshold": (u32, [vp]), "mko_k": (u32, [vp]),
        "mko_h": (u32, [vp]), "mko_fp_bits": (u32, [vp]),
        "mko_bloom_bytes": (u64, [vp]), "mko_bloom": (vp, [vp]),
        "mko_sketch_size": (vp, [vp]), "mko_genome_size": (vp, [vp]),
        "mko_column": (vp, [vp, u32]),
        "mko_mantis": (u32, [vp, u64]),
        "mko_sketch": (u32, [vp, cp, u64, vp, vp]),
        "mko_sketch_solid": (None, [vp, cp, u64, vp]),
        "mko_check_bloom": (C.c_int, [vp, u64]),
        "mko_insert_sequences": (None, [vp, vp, vp, u32]),
        "mko_insert_sequence": (None, [vp, vp, u64]),
        "mko_query_sequences": (None, [vp, vp, vp, u32, vp]),
        "mko_query_sequence": (u32, [vp, cp, u64, vp]),
        "mko_filter_results": (u32, [vp, vp, u32, u32, C.c_double, vp]),
        "mko_format_query_line": (C.c_size_t, [cp, vp, u32, cp]),
        "mko_serial_size": (u64, [vp]), "mko_serialize": (None, [vp, vp]),
        "mko_deserialize": (vp, [vp, u64]),
        "mko_exact_genome_set": (u64, [cp, u64, u32, C.POINTER(vp)]),
        "mko_exact_query": (None, [vp, u64, cp, u64, u32, C.POINTER(u64), C.POINTER(u64)]),
        "mko_free": (None, [vp]),
        "mko_poke_sizes": (None, [vp, u32, vp, vp]),
        "mko_revhash64": (u64, [u64]), "mko_unrevhash64": (u64, [u64]),
        "mko_universal_hash": (u64, [u64, u32]), "mko_str2num": (u64, [cp, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


_L = None


def lib():
    global _L
    if _L is None:
        _L = _load()
    return _L


def _seq_arrays(seqs):
    n = len(seqs)
    ptrs = (C.c_char_p * n)(*seqs)
    lens = (C.c_uint64 * n)(*[len(s) for s in seqs])
    return ptrs, lens


class OracleMiekki:
    """CPU restatement of the reference's Miekki object (hot path only)."""

    def __init__(self, k=31, h=17, fp_bits=8, bloom_log2=33, threshold=200, _handle=None):
        self._L = lib()
        self._h = _handle or self._L.mko_create(k, h, fp_bits, bloom_log2, threshold)
        if not self._h:
            raise ValueError("not implemented")           # Miekki.cpp:235-237

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.mko_destroy(self._h)
            self._h = None

    # -- metadata
    @property
    def index_size(self): return self._L.mko_index_size(self._h)
    @property
    def threshold(self): return self._L.mko_threshold(self._h)
    @property
    def kmer_size(self): return self._L.mko_k(self._h)
    @property
    def number_minimizer_log2(self): return self._L.mko_h(self._h)
    @property
    def number_bit_minimizer(self): return self._L.mko_fp_bits(self._h)
    @property
    def W(self): return self.number_bit_minimizer // 8
    @property
    def P(self): return 1 << self.number_minimizer_log2

    @property
    def sketch_size(self):
        G = self.index_size
        return np.ctypeslib.as_array(C.cast(self._L.mko_sketch_size(self._h), C.POINTER(C.c_uint32)), (G,)).copy() if G else np.zeros(0, np.uint32)

    @property
    def genome_size(self):
        G = self.index_size
        return np.ctypeslib.as_array(C.cast(self._L.mko_genome_size(self._h), C.POINTER(C.c_uint64)), (G,)).copy() if G else np.zeros(0, np.uint64)

    @property
    def bloom(self):
        n = self._L.mko_bloom_bytes(self._h)
        return np.ctypeslib.as_array(C.cast(self._L.mko_bloom(self._h), C.POINTER(C.c_uint8)), (n,))

    def columns(self):
        """[P, G*W] uint8, partition-major exactly as the reference stores/dumps it."""
        G, W = self.index_size, self.W
        out = np.empty((self.P, G * W), np.uint8)
        for p in range(self.P):
            out[p] = np.ctypeslib.as_array(C.cast(self._L.mko_column(self._h, p), C.POINTER(C.c_uint8)), (G * W,))
        return out

    # -- reference API
    def minhash_sketch_partition(self, seq: bytes):
        P = self.P
        fp = np.empty(P, np.uint16); hs = np.empty(P, np.uint64)
        act = self._L.mko_sketch(self._h, seq, len(seq), fp.ctypes.data, hs.ctypes.data)
        return fp, hs, act

    def minhash_sketch_partition_solid_kmers(self, seq: bytes):
        fp = np.empty(self.P, np.uint16)
        self._L.mko_sketch_solid(self._h, seq, len(seq), fp.ctypes.data)
        return fp

    def insert_sequences(self, seqs):
        ptrs, lens = _seq_arrays(seqs)
        self._L.mko_insert_sequences(self._h, ptrs, lens, len(seqs))

    def insert_sequence(self, seq: bytes):
        seq = bytes(seq)
        self._L.mko_insert_sequence(self._h, seq, len(seq))

    def query_sequences(self, seqs):
        ptrs, lens = _seq_arrays(seqs)
        out = np.zeros((len(seqs), self.index_size), np.uint32)
        self._L.mko_query_sequences(self._h, ptrs, lens, len(seqs), out.ctypes.data)
        return out

    def query_sequence(self, seq: bytes):
        out = np.zeros(self.index_size, np.uint32)
        act = self._L.mko_query_sequence(self._h, seq, len(seq), out.ctypes.data)
        return out, act

    def filter_results(self, scores_row, nresults, min_score, min_intersection):
        row = np.ascontiguousarray(scores_row, np.uint32)
        buf = (Hit * max(nresults, 1))()
        n = self._L.mko_filter_results(self._h, row.ctypes.data, nresults, min_score,
                                       float(min_intersection), buf)
        return [(buf[i].genome, buf[i].matches, buf[i].jaccard, buf[i].intersection) for i in range(n)]

    def format_query_line(self, name: bytes, hits):
        buf = (Hit * max(len(hits), 1))()
        for i, (g, m, j, x) in enumerate(hits):
            buf[i] = Hit(g, m, j, x)
        out = C.create_string_buffer(len(name) + 8 + 96 * len(hits))
        n = self._L.mko_format_query_line(name, buf, len(hits), out)
        return out.raw[:n]

    # -- persistence (uncompressed stream of SURVEY row P)
    def serialize(self) -> np.ndarray:
        n = self._L.mko_serial_size(self._h)
        buf = np.empty(n, np.uint8)
        self._L.mko_serialize(self._h, buf.ctypes.data)
        return buf

    @classmethod
    def deserialize(cls, raw: np.ndarray):
        raw = np.ascontiguousarray(raw, np.uint8)
        h = lib().mko_deserialize(raw.ctypes.data, raw.size)
        if not h:
            raise ValueError("malformed index stream")
        return cls(_handle=h)

    # -- sizes the reference stores after a direct poke (for synthetic filter tests)
    def poke_sizes(self, sketch_size, genome_size):
        ss = np.ascontiguousarray(sketch_size, np.uint32)
        gs = np.ascontiguousarray(genome_size, np.uint64)
        self._L.mko_poke_sizes(self._h, ss.size, ss.ctypes.data, gs.ctypes.data)


def exact_genome_set(fasta: bytes, k: int) -> np.ndarray:
    L = lib()
    p = C.c_void_p()
    n = L.mko_exact_genome_set(fasta, len(fasta), k, C.byref(p))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), (n,)).copy() if n else np.zeros(0, np.uint64)
    L.mko_free(p)
    return arr


def exact_query(kset: np.ndarray, seq: bytes, k: int):
    L = lib()
    kset = np.ascontiguousarray(kset, np.uint64)
    a, b = C.c_uint64(), C.c_uint64()
    L.mko_exact_query(kset.ctypes.data, kset.size, seq, len(seq), k, C.byref(a), C.byref(b))
    return a.value, b.value
