"""Multi-GPU layer: genome-sharded index, one gather of per-query heap entrants.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests).  Rank r owns the contiguous genome range
shard_range(G, r, world); every rank scans all queries against its shard and
emits, per query, the candidates that passed filter_results' two thresholds
(Miekki.cpp:381-384) in ascending genome id.  Because shards are contiguous and
ordered by rank, concatenating the gathered rows in rank order reproduces exactly
the sequence in which the reference's filter_results meets the genomes
(Miekki.cpp:379), so the heap on rank 0 (mk_filter_candidates) is the reference's.
There is no other data-path collective.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import lib as L
from .shard import shard_range  # noqa: F401  (re-exported)

HIT_BYTES = 24        # sizeof(mk_hit)
HIT_DTYPE = np.dtype([("genome", "<u4"), ("matches", "<u4"), ("jaccard", "<f8"), ("intersection", "<f8")])
assert HIT_DTYPE.itemsize == HIT_BYTES




def gather_rows(count: torch.Tensor, cand: torch.Tensor, dst: int = 0, group=None, out=None):
    """The single exchange step.  count: int32 [nq]; cand: uint8 [nq*cap*24].
    Returns tensors (counts [world, nq], cands [world, nq*cap*24]) on dst -- on the device
    the inputs live on, rank-major, i.e. the layout mk_merge_entrants takes -- and
    (None, None) elsewhere.  `out` = a (counts, cands) pair to receive into (reused by
    bench.py from step to step)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gc = gd = big_c = big_d = None
    if rank == dst:
        big_c, big_d = out if out is not None else (count.new_empty((world, count.numel())),
                                                    cand.new_empty((world, cand.numel())))
        gc = [big_c[r] for r in range(world)]
        gd = [big_d[r] for r in range(world)]
    dist.gather(count, gc, dst=dst, group=group)
    dist.gather(cand, gd, dst=dst, group=group)
    return big_c, big_d


ROW_WORDS = lambda cap: cap + 1     # 64-bit words of one exchange row: count, then cap x (genome | matches << 32)


def gather_compact(rows: torch.Tensor, dst: int = 0, group=None, out=None):
    """The single exchange step in its 8-byte form.  rows: int64
    [nq * (cap + 1)] as mk_qset_run_compact wrote them.  ONE collective; returns the
    tensor [world, nq * (cap + 1)] on dst (rank-major = shard order = genome order, the
    layout mk_merge_compact takes) and None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    big = lst = None
    if rank == dst:
        big = out if out is not None else rows.new_empty((world, rows.numel()))
        lst = [big[r] for r in range(world)]
    dist.gather(rows, lst, dst=dst, group=group)
    return big


def gather_sizes(ss: np.ndarray, gs: np.ndarray, device=None, group=None):
    """All ranks' (sketch_size u32[], genome_size u64[]) concatenated in rank order = genome id
    order, on every rank.  Two small collectives, once after the build."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return np.ascontiguousarray(ss, np.uint32), np.ascontiguousarray(gs, np.uint64)
    n = torch.tensor([len(ss)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(t.item()) for t in ns]
    m = max(ns + [1])
    pack = torch.zeros(2 * m, dtype=torch.int64)
    pack[:len(ss)] = torch.from_numpy(np.asarray(ss).astype(np.int64))
    pack[m:m + len(gs)] = torch.from_numpy(np.ascontiguousarray(gs, np.uint64).view(np.int64))
    pack = pack.to(device) if device is not None else pack
    parts = [torch.zeros_like(pack) for _ in range(world)]
    dist.all_gather(parts, pack, group=group)
    parts = [p.cpu().numpy() for p in parts]
    ss_all = np.concatenate([parts[r][:ns[r]] for r in range(world)]).astype(np.uint32)
    gs_all = np.concatenate([parts[r][m:m + ns[r]] for r in range(world)]).view(np.uint64)
    return np.ascontiguousarray(ss_all), np.ascontiguousarray(gs_all)


def share_sizes(ix, device=None, group=None):
    """Once after the build: gather_sizes of every rank's index, handed to the library
    (mk_merge_set_sizes) so that the merging rank can recompute jaccard and intersection
    of a gathered (genome, matches) record in the reference's double operations.  Ranks
    hold contiguous id ranges in rank order starting at id 0.  Returns the global arrays."""
    lib = L.load_library()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ss, gs = gather_sizes(ix.sketch_size, ix.genome_size, device, group)
    base = 0 if world > 1 else ix._p.genome_id_base
    L.check(lib.mk_merge_set_sizes(ix._h, gs.ctypes.data, ss.ctypes.data, len(ss), base))
    return ss, gs


def merge_compact_on_device(ix, rows: torch.Tensor, nq: int, cap: int, nresults: int, out=None):
    """Rank-0 merge of gathered exchange rows on the GPU (K6b, mk_merge_compact): rows
    int64 [world, nq * (cap + 1)] resident on this rank's GPU.  Returns (hits uint8
    [nq, nresults * 24], nhits int32 [nq]); nhits == MK_MERGE_OVERFLOW marks a query
    some shard overflowed for."""
    lib = L.load_library()
    assert rows.is_cuda and rows.is_contiguous()
    world = rows.shape[0] if rows.dim() == 2 else 1
    hits, nhits = out if out is not None else (
        torch.empty((nq, max(nresults, 1) * HIT_BYTES), dtype=torch.uint8, device=rows.device),
        torch.empty(nq, dtype=torch.int32, device=rows.device))
    L.check(lib.mk_merge_compact(ix._h, rows.data_ptr(), world, nq, cap, nresults, hits.data_ptr(), nhits.data_ptr()))
    return hits, nhits


def merge_compact_host(rows: np.ndarray, nq: int, cap: int, nresults: int, ss: np.ndarray, gs: np.ndarray):
    """Host restatement of mk_merge_compact for tests and bench self-checks: rows uint64
    [world, nq * (cap + 1)]; sizes indexed by global genome id.  Returns (hits, overflow)."""
    lib = L.load_library()
    world = rows.shape[0]
    r3 = rows.reshape(world, nq, cap + 1)
    counts = (r3[:, :, 0] & 0xFFFFFFFF).astype(np.int64)
    overflow = (counts > cap).any(axis=0)
    out = []
    buf = np.empty(world * cap, HIT_DTYPE)
    res = np.empty(max(nresults, 1), HIT_DTYPE)
    for q in range(nq):
        n = 0
        for r in range(world):
            m = int(min(counts[r, q], cap))
            if m:
                rec = r3[r, q, 1:1 + m]
                g = (rec & 0xFFFFFFFF).astype(np.uint32)
                mt = (rec >> 32).astype(np.uint32)
                with np.errstate(divide="ignore", invalid="ignore"):
                    jac = mt.astype(np.float64) / ss[g].astype(np.float64)
                    inter = jac * gs[g].astype(np.float64)
                buf["genome"][n:n + m] = g; buf["matches"][n:n + m] = mt
                buf["jaccard"][n:n + m] = jac; buf["intersection"][n:n + m] = inter
                n += m
        k = lib.mk_filter_candidates(buf.ctypes.data_as(C.c_void_p), n, nresults, res.ctypes.data_as(C.c_void_p))
        out.append(res[:k].copy())
    return out, overflow


def gather_candidates(count: torch.Tensor, cand: torch.Tensor, dst: int = 0, group=None):
    """gather_rows, as numpy arrays on the host (the CPU-side merge and the tests)."""
    gc, gd = gather_rows(count, cand, dst, group)
    if gc is None:
        return None, None
    return gc.cpu().numpy(), gd.cpu().numpy()


def merge_candidates(counts: np.ndarray, cands: np.ndarray, cap: int, nresults: int):
    """Rank-0 merge: per query concatenate the shards' rows in rank order and run
    the reference heap.  Returns (hits per query as structured arrays, overflow mask:
    queries for which some shard had more candidates than `cap`)."""
    lib = L.load_library()
    world, nq = counts.shape
    rows = np.ascontiguousarray(cands).view(np.uint8).reshape(world, nq, cap * HIT_BYTES)
    overflow = (counts > cap).any(axis=0)
    out = []
    buf = np.empty(world * cap, HIT_DTYPE)
    res = np.empty(max(nresults, 1), HIT_DTYPE)
    for q in range(nq):
        n = 0
        for r in range(world):
            m = int(min(counts[r, q], cap))
            if m:
                buf[n:n + m] = rows[r, q, :m * HIT_BYTES].view(HIT_DTYPE)
                n += m
        k = lib.mk_filter_candidates(buf.ctypes.data_as(C.c_void_p), n, nresults, res.ctypes.data_as(C.c_void_p))
        out.append(res[:k].copy())
    return out, overflow


MERGE_OVERFLOW = 0xFFFFFFFF


def merge_on_device(ix, counts: torch.Tensor, cands: torch.Tensor, cap: int, nresults: int):
    """Rank-0 merge on the GPU (K6b, mk_merge_entrants): counts int32 [world, nq] and cands
    uint8 [world, nq*cap*24] are the gathered rows, resident on this rank's GPU.  Returns
    device tensors (hits uint8 [nq, nresults*24], nhits int32 [nq]); nhits == -1
    (MK_MERGE_OVERFLOW) marks a query some shard overflowed for.  The caller must have
    ordered the producers of counts/cands before the library's stream (e.g.
    torch.cuda.current_stream().synchronize())."""
    lib = L.load_library()
    world, nq = counts.shape
    assert counts.is_cuda and cands.is_cuda and counts.is_contiguous() and cands.is_contiguous()
    hits = torch.empty((nq, max(nresults, 1) * HIT_BYTES), dtype=torch.uint8, device=counts.device)
    nhits = torch.empty(nq, dtype=torch.int32, device=counts.device)
    L.check(lib.mk_merge_entrants(ix._h, counts.data_ptr(), cands.data_ptr(), world, nq, cap, nresults,
                                  hits.data_ptr(), nhits.data_ptr()))
    return hits, nhits


def merge_bloom_first_writer(local: np.ndarray, device=None, group=None) -> np.ndarray:
    """Global Bloom filter of a genome-sharded build, byte-exact.

    The reference has ONE filter for the whole collection and a cell keeps the bit
    of its FIRST inserter in genome order (Miekki.cpp:125-129).  With contiguous
    shards ordered by rank the first inserter lives on the lowest rank whose cell is
    non-zero, so a MIN all-reduce over (rank << 8 | byte), with 0xFFFF for empty
    cells, reproduces the single-process bytes.  One collective, once per build."""
    rank = dist.get_rank(group)
    t = torch.from_numpy(local.astype(np.int32))
    t = torch.where(t == 0, torch.full_like(t, 0xFFFF), t | (rank << 8))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    t = t.cpu()
    return torch.where(t == 0xFFFF, torch.zeros_like(t), t & 0xFF).to(torch.uint8).numpy()


def sync_bloom(ix, device=None, group=None):
    """Make every rank's Bloom gate the global one (call once after the build).  With a
    CUDA `device` the cells never leave the GPUs: they are exported to a device tensor,
    keyed, MIN-all-reduced over RCCL and imported back."""
    lib = L.load_library()
    nb = ix.bloom_size // 8
    if nb == 0:
        return
    # only the cells a 2k-bit k-mer can reach are ever non-zero
    reach = min(nb, (((1 << (2 * ix.kmer_size)) - 1 + 1023) >> (ix.bloom_size_log2 + 3)) + 1)
    if device is not None and torch.device(device).type == "cuda":
        rank = dist.get_rank(group)
        cells = torch.empty(reach, dtype=torch.uint8, device=device)
        L.check(lib.mk_index_export_bloom_device(ix._h, 0, reach, cells.data_ptr()))
        step = 16 << 20                                # bounded temporaries: 64 MiB of int32 keys at a time
        for o in range(0, reach, step):
            c = cells[o:o + step]
            t = c.to(torch.int32)
            t = torch.where(t == 0, torch.full_like(t, 0xFFFF), t | (rank << 8))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            c.copy_(torch.where(t == 0xFFFF, torch.zeros_like(t), t & 0xFF).to(torch.uint8))
        torch.cuda.synchronize()
        L.check(lib.mk_index_import_bloom_device(ix._h, 0, reach, cells.data_ptr()))
        return
    local = np.empty(reach, np.uint8)
    L.check(lib.mk_index_export_bloom(ix._h, 0, reach, local.ctypes.data))
    merged = merge_bloom_first_writer(local, device, group)
    L.check(lib.mk_index_import_bloom(ix._h, 0, reach, merged.ctypes.data))
