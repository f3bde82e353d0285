"""Genome sharding rule of the multi-GPU path (no torch, no GPU: importable anywhere)."""


def shard_range(n_genomes: int, rank: int, world: int):
    """Contiguous, ordered genome-id range of a rank (sizes differ by at most one)."""
    base, rem = divmod(n_genomes, world)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)


def entrant_cap(nresults: int, shard_genomes: int) -> int:
    """Entrant slots per query of one shard's exchange row (host/multi_gpu.hpp: entrant_cap, the same rule).
    The entrants of filter_results' heap (Miekki.cpp:387: everything not below the current minimum) among m
    candidates in genome order number about N (1 + ln(m / N)), more when scores tie (ties enter); measured at
    -h 20, N = 10 (profiles/r4_entrant_rows.txt): 12,500-genome shards 64 +- 7 (max 95 of 8,192
    queries), 50,000: 77 +- 8 (0.6 % over 96 slots, max 110), 100,000: 83 +- 8 (4.8 % over 96, max 114).  N (3 + ln(G / N)) rounded up to 32 slots keeps
    five deviations of room; rows that overflow anyway are run again with wide rows."""
    import math
    n = max(int(nresults), 1)
    want = n * (3.0 + math.log(max(shard_genomes / n, 1.0)))
    return max(64, 32 * int(math.ceil(want / 32.0)))
