"""Multi-GPU layer: genome-sharded index, one gather of per-query candidates.

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

HIT_BYTES = 24        # sizeof(mk_hit)
HIT_DTYPE = np.dtype([("genome", "<u4"), ("matches", "<u4"), ("jaccard", "<f8"), ("intersection", "<f8")])
assert HIT_DTYPE.itemsize == HIT_BYTES


def shard_range(n_genomes: int, rank: int, world: int):
    """Contiguous, ordered genome-id range of a rank (sizes differ by at most one)."""
    base, rem = divmod(n_genomes, world)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)


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
    """Make every rank's Bloom gate the global one (call once after the build)."""
    lib = L.load_library()
    nb = ix.bloom_size // 8
    if nb == 0:
        return
    # only the cells a 2k-bit k-mer can reach are ever non-zero
    reach = min(nb, (((1 << (2 * ix.kmer_size)) - 1 + 1023) >> (ix.bloom_size_log2 + 3)) + 1)
    local = np.empty(reach, np.uint8)
    L.check(lib.mk_index_export_bloom(ix._h, 0, reach, local.ctypes.data))
    merged = merge_bloom_first_writer(local, device, group)
    L.check(lib.mk_index_import_bloom(ix._h, 0, reach, merged.ctypes.data))
