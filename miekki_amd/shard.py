"""Genome sharding rule of the multi-GPU path (no torch, no GPU: importable anywhere)."""


def shard_range(n_genomes: int, rank: int, world: int):
    """Contiguous, ordered genome-id range of a rank (sizes differ by at most one)."""
    base, rem = divmod(n_genomes, world)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)
