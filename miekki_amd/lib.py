"""ctypes binding of include/miekki_hip.h (libmiekki_hip.so)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmiekki_hip.so")


class MiekkiHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libmiekki_hip: status {status}: {msg}")
        self.status = status


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("h", C.c_uint32), ("fp_bits", C.c_uint32), ("bloom_log2", C.c_uint32),
                ("threshold", C.c_uint32), ("device", C.c_int32), ("genome_id_base", C.c_uint32),
                ("reserved", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("genome", C.c_uint32), ("matches", C.c_uint32), ("jaccard", C.c_double),
                ("intersection", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("sketch_ms", C.c_double), ("scan_ms", C.c_double), ("filter_ms", C.c_double),
                ("scan_launches", C.c_uint64), ("comparisons", C.c_uint64), ("active_partitions", C.c_uint64),
                ("scan_algo_bytes", C.c_uint64), ("build_sketch_ms", C.c_double),
                ("build_finalize_ms", C.c_double), ("build_kmers", C.c_uint64), ("build_genomes", C.c_uint64),
                ("scan_slab_launches", C.c_uint64), ("peer_copies", C.c_uint64), ("staged_copies", C.c_uint64),
                ("peer_copy_bytes", C.c_uint64), ("staged_copy_bytes", C.c_uint64)]


class PackedSeq(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("except_", C.c_void_p), ("len", C.c_uint64), ("head", C.c_char * 32)]


vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
PP = C.POINTER

# name -> (restype, argtypes): every symbol include/miekki_hip.h declares
SIGNATURES = {
    "mk_last_error": (C.c_char_p, []),
    "mk_abi_version": (u32, []),
    "mk_create": (i32, [PP(Params), PP(vp)]),
    "mk_destroy": (None, [vp]),
    "mk_reserve": (i32, [vp, u32]),
    "mk_index_compress": (i32, [vp, PP(u64), PP(u64)]),
    "mk_index_decompress": (i32, [vp]),
    "mk_index_size": (u32, [vp]),
    "mk_get_params": (i32, [vp, PP(Params)]),
    "mk_get_stats": (i32, [vp, PP(Stats)]),
    "mk_reset_stats": (i32, [vp]),
    "mk_probe_stream_read": (i32, [vp, u32, PP(C.c_double), PP(u64)]),
    "mk_probe_synth_genomes": (i32, [vp, u64, u32, u64, vp]),
    "mk_index_append": (i32, [vp, vp, vp, u32]),
    "mk_index_insert_sequence": (i32, [vp, vp, u64]),
    "mk_index_append_packed": (i32, [vp, vp, u32]),
    "mk_pack_code_words": (u64, [u64]),
    "mk_pack_except_words": (u64, [u64]),
    "mk_pack_append": (i32, [vp, vp, u64, vp, u64]),
    "mk_host_alloc": (i32, [vp, u64, PP(vp)]),
    "mk_host_free": (None, [vp, vp]),
    "mk_index_append_synthetic": (i32, [vp, u64, u32, u64]),
    "mk_index_append_synthetic_strains": (i32, [vp, u64, u32, u64, u32, u32]),
    "mk_index_export_columns": (i32, [vp, u32, u32, vp]),
    "mk_index_export_genomes": (i32, [vp, vp, u32, vp]),
    "mk_index_export_sizes": (i32, [vp, vp, vp]),
    "mk_index_export_bloom": (i32, [vp, u64, u64, vp]),
    "mk_index_import_begin": (i32, [vp, u32]),
    "mk_index_import_columns": (i32, [vp, u32, u32, vp]),
    "mk_index_import_columns_huffman": (i32, [vp, u32, u32, vp, u64, vp, u32, vp, u32, vp, vp]),
    "mk_index_import_sizes": (i32, [vp, vp, vp]),
    "mk_index_import_bloom": (i32, [vp, u64, u64, vp]),
    "mk_query_scores": (i32, [vp, vp, vp, u32, vp]),
    "mk_query": (i32, [vp, vp, vp, u32, u32, u32, C.c_double, vp, vp, vp]),
    "mk_filter_candidates": (u32, [vp, u32, u32, vp]),
    "mk_merge_entrants": (i32, [vp, vp, vp, u32, u32, u32, u32, vp, vp]),
    "mk_qset_upload": (i32, [vp, vp, vp, u32, PP(vp)]),
    "mk_qset_synthetic": (i32, [vp, u64, u32, u64, u64, u64, PP(vp)]),
    "mk_qset_free": (None, [vp, vp]),
    "mk_qset_run": (i32, [vp, vp, u32, u32, C.c_double, u32, vp, vp]),
    "mk_qset_run_compact": (i32, [vp, vp, u32, u32, C.c_double, u32, vp]),
    "mk_qset_invalidate": (i32, [vp, vp]),
    "mk_merge_set_sizes": (i32, [vp, vp, vp, u32, u32]),
    "mk_merge_get_sizes": (i32, [vp, vp, vp, u32]),
    "mk_merge_compact": (i32, [vp, vp, u32, u32, u32, u32, vp, vp]),
    "mk_device_count": (i32, []),
    "mk_set_genome_id_base": (i32, [vp, u32]),
    "mk_dev_alloc": (i32, [vp, u64, PP(vp)]),
    "mk_dev_free": (None, [vp, vp]),
    "mk_dev_upload": (i32, [vp, vp, vp, u64]),
    "mk_dev_download": (i32, [vp, vp, vp, u64]),
    "mk_dev_copy": (i32, [vp, vp, vp, vp, u64]),
    "mk_index_export_bloom_device": (i32, [vp, u64, u64, vp]),
    "mk_index_import_bloom_device": (i32, [vp, u64, u64, vp]),
    "mk_index_merge_bloom_device": (i32, [vp, u64, u64, vp]),
    "mk_bloom_reachable_bytes": (u64, [vp]),
    "mk_comm_unique_id": (i32, [vp]),
    "mk_comm_create": (i32, [vp, i32, i32, vp, PP(vp)]),
    "mk_comm_destroy": (None, [vp]),
    "mk_comm_rank": (i32, [vp]),
    "mk_comm_world": (i32, [vp]),
    "mk_comm_gather": (i32, [vp, vp, u64, vp, i32]),
    "mk_comm_allgather": (i32, [vp, vp, u64, vp]),
    "mk_comm_broadcast": (i32, [vp, vp, u64, i32]),
    "mk_comm_allreduce_max_f64": (i32, [vp, vp, u32]),
    "mk_comm_barrier": (i32, [vp]),
    "mk_comm_gather_rows": (i32, [vp, vp, u64, vp, i32]),
    "mk_comm_sync_bloom": (i32, [vp]),
    "mk_comm_share_sizes": (i32, [vp, PP(u32), PP(u32)]),
    "mk_qset_run_compact_gather": (i32, [vp, vp, vp, u32, u32, C.c_double, u32, vp, vp, i32]),
    "mk_qset_scores": (i32, [vp, vp, u32, u32, vp]),
    "mk_qset_active": (i32, [vp, vp, vp]),
    "mk_sync": (i32, [vp]),
    "mk_gz_inflate": (i32, [vp, vp, vp, u32, vp, vp, vp, vp]),
    "mk_gz_unpack": (i32, [vp, vp, vp, u32, PP(vp)]),
    "mk_gz_open": (i32, [vp, vp, u32, PP(vp)]),
    "mk_gz_stage": (vp, [vp, PP(u64)]),
    "mk_gz_put": (i32, [vp, u32, u64, vp, u64, C.c_int]),
    "mk_gz_layout": (i32, [vp, vp]),
    "mk_gz_put_span": (i32, [vp, u32, u32, vp, u64, C.c_int]),
    "mk_gz_run": (i32, [vp]),
    "mk_gz_sequence": (i32, [vp, u32, PP(u64), PP(C.c_int32)]),
    "mk_index_append_gz": (i32, [vp, vp, vp, u32]),
    "mk_gz_free": (None, [vp]),
    "mk_gz_trim": (None, [vp]),
    "mk_exact": (i32, [vp, vp, vp, u32, vp, vp, u32, vp, vp]),
    "mk_exact_load_genome": (i32, [vp, vp, vp, u32]),
    "mk_exact_query": (i32, [vp, vp, vp, u32, vp, vp]),
}

_lib = None


def library_path() -> str:
    return _LIB_PATH


def _share_torch_hip_runtime():
    """Keep ONE HIP runtime in the process.  The PyTorch wheel bundles its own
    libamdhip64.so (SONAME libamdhip64.so.7) which its libraries request by FILE name,
    so a system runtime loaded first for us is not reused by a later `import torch`:
    the process would then hold two runtimes and the second one finds no GPU.  Loading
    torch's copy first (when torch is installed) makes our library bind to it by
    SONAME, whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                     # fall back to the system runtime


def load_library():
    """Load libmiekki_hip.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise MiekkiHipError(-3, f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                     "or `make -C miekki_amd/csrc lib` (there is no CPU fallback)")
        _share_torch_hip_runtime()
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(status: int):
    if status != 0:
        raise MiekkiHipError(status, load_library().mk_last_error().decode(errors="replace"))


def gz_inflate(ctx_handle, blobs, rooms):
    """mk_gz_inflate: whole gzip files (bytes) -> (texts, statuses); a text is None where the status is not MK_GZ_OK."""
    import numpy as np
    lib = load_library()
    n = len(blobs)
    ptrs, lens = seq_arrays(blobs)
    outs = [np.empty(max(int(r), 1), np.uint8) for r in rooms]
    out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    room = (C.c_uint64 * n)(*[int(r) for r in rooms])
    got = (C.c_uint64 * n)()
    status = (C.c_int32 * n)()
    check(lib.mk_gz_inflate(ctx_handle, ptrs, lens, n, out_ptrs, room, got, status))
    return [bytes(outs[i][:got[i]]) if status[i] == 0 else None for i in range(n)], list(status)


def seq_arrays(seqs):
    n = len(seqs)
    return (C.c_char_p * n)(*seqs), (C.c_uint64 * n)(*[len(s) for s in seqs])
