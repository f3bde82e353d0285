"""N > 1 path on CPU: world_size-2 gloo run of the genome-sharded merge.

Each rank holds an ORACLE index of its genome shard (the HIP path needs a GPU;
what is under test here is the sharding rule, the single gather and the rank-0
merge, which are device independent).  The merged hits must equal the
unsharded oracle's filter_results, including heap tie behaviour."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from miekki_amd import distributed as mkd

K, H, THR = 21, 10, 0
CAP, NRES = 8, 5


def make_inputs():
    base = synth.genome_bases(777, 0, 6000)
    genomes = [synth.genome_bases(500 + g, 0, 4000) for g in range(9)]
    genomes[3] = base[:4000]; genomes[7] = base[:4000]        # identical genomes -> equal intersections (ties)
    genomes[5] = base[1000:5000]
    queries = [base[200:1800], genomes[0][100:900], genomes[8][:3000], synth.genome_bases(9, 0, 500), base[:K]]
    return genomes, queries


def shard_rows(o, queries, g0, g1, min_score, min_inter):
    """The shard's rows in the 8-byte exchange form (what mk_qset_run_compact writes): per query
    [count, genome | matches << 32 ...]; all thresholded candidates = a superset of the entrants."""
    scores = o.query_sequences(queries)
    ss, gs = o.sketch_size.astype(np.float64), o.genome_size.astype(np.float64)
    rows = np.zeros((len(queries), CAP + 1), np.uint64)
    for q in range(len(queries)):
        n = 0
        for g in range(g1 - g0):                               # ascending genome id
            s = scores[q, g]
            jac = float(s) / ss[g]; inter = jac * gs[g]
            if s < min_score or inter < min_inter:
                continue
            if n < CAP:
                rows[q, 1 + n] = (g + g0) | (int(s) << 32)
            n += 1
        rows[q, 0] = n
    return rows


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genomes, queries = make_inputs()
    g0, g1 = mkd.shard_range(len(genomes), rank, world)
    from oracle import oracle as orc
    o = orc.OracleMiekki(K, H, 8, 32, THR)
    o.insert_sequences(genomes[g0:g1])
    # global Bloom gate from the shards' filters, byte-exact (first writer = lowest rank)
    merged = mkd.merge_bloom_first_writer(o.bloom[:1 << 20].copy())
    o.bloom[:1 << 20] = merged
    rows = shard_rows(o, queries, g0, g1, 1, 0.0)
    ss_all, gs_all = mkd.gather_sizes(o.sketch_size, o.genome_size)         # once, after the build
    big = mkd.gather_compact(torch.from_numpy(rows.view(np.int64).reshape(-1).copy()))   # the ONE exchange
    if rank == 0:
        ret["bloom"] = merged.tobytes()
        ret["gather_bytes_per_rank"] = rows.nbytes
        hits, overflow = mkd.merge_compact_host(big.numpy().view(np.uint64), len(queries), CAP, NRES, ss_all, gs_all)
        ret["hits"] = [[(int(h["genome"]), int(h["matches"]), float(h["intersection"])) for h in row] for row in hits]
        ret["overflow"] = overflow.tolist()
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_are_contiguous_and_ordered():
    for n, w in ((100_000, 8), (9, 2), (7, 8), (0, 4)):
        r = [mkd.shard_range(n, i, w) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_gloo_merge_equals_unsharded_reference():
    from oracle import oracle as orc
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(2, free_port(), ret), nprocs=2, join=True)
    genomes, queries = make_inputs()
    o = orc.OracleMiekki(K, H, 8, 32, THR)
    o.insert_sequences(genomes)
    scores = o.query_sequences(queries)
    assert not any(ret["overflow"])
    assert ret["gather_bytes_per_rank"] == len(queries) * (CAP + 1) * 8     # 8-byte records + one header word
    # k=21, b=32: reachable cells < 2^(42-35) -- the first 2^20 bytes cover them all
    assert o.bloom[1 << 20:].max() == 0
    assert ret["bloom"] == o.bloom[:1 << 20].tobytes()
    for q in range(len(queries)):
        want = o.filter_results(scores[q], NRES, 1, 0.0)
        got = ret["hits"][q]
        assert [(g, m) for g, m, _ in got] == [(w[0], w[1]) for w in want], q
        np.testing.assert_allclose([x for _, _, x in got], [w[3] for w in want], rtol=1e-12)
