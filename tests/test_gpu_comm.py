"""Everything of the N > 1 path that ONE GPU can execute (RCCL had never run on device tensors).

A process group of one on the nccl backend (= RCCL) and a communicator of one through the C ABI (mk_comm_*: ncclGather,
ncclAllReduce, ncclAllGather, grouped ncclSend / ncclRecv called from libmiekki_hip.so): RCCL initialises, every
collective of the sharded path runs on DEVICE memory, the library's streams and torch's hand over to each other, and the
results are those of the plain single-context calls.  The multi-rank semantics (who contributes what) are covered on the
CPU by tests/test_distributed_cpu.py (gloo, world size 2) and in rehearsal by tests/test_gpu_bench.py / test_gpu_cli.py.
"""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, L_, K, H = 700, 60_000, 31, 16


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def index():
    import miekki_amd
    ix = miekki_amd.Miekki(K, H, 8, 33, 20)
    seqs = [synth.genome_bases(g, 0, L_) for g in range(G)]
    for i in range(0, G, 64):
        ix.insert_sequences(seqs[i:i + 64])
    yield ix
    ix.close()


def make_queries(n):
    return [synth.genome_bases(*synth.query_origin(q, G, L_, 1000), 1000) for q in range(n)]


def plain_rows(ix, qset, nq, cap):
    import torch
    from miekki_amd import lib as L
    rows = torch.zeros(nq * (cap + 1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    L.check(ix._lib.mk_qset_run_compact(ix._h, qset, 10, 4, 1.0, cap, rows.data_ptr()))
    L.check(ix._lib.mk_sync(ix._h))
    return rows


def test_torch_nccl_collectives_on_device_tensors(pg, index):
    """miekki_amd.distributed on the nccl backend with device tensors: sync_bloom (the device path: keyed MIN all-reduce),
    share_sizes, gather_compact, merge_compact_on_device -- at world size 1 each must be the identity."""
    import torch
    from miekki_amd import distributed as mkd
    from miekki_amd import lib as L
    ix, lib = index, index._lib
    dev = torch.device("cuda", 0)
    assert pg.get_backend() == "nccl" and pg.get_world_size() == 1
    reach = lib.mk_bloom_reachable_bytes(ix._h)
    before = np.empty(reach, np.uint8)
    L.check(lib.mk_index_export_bloom(ix._h, 0, reach, before.ctypes.data))
    assert before.any()
    mkd.sync_bloom(ix, device=dev)                               # export -> all_reduce(MIN) on the GPU -> import
    after = np.empty(reach, np.uint8)
    L.check(lib.mk_index_export_bloom(ix._h, 0, reach, after.ctypes.data))
    np.testing.assert_array_equal(before, after)
    ss, gs = mkd.share_sizes(ix, device=dev)
    np.testing.assert_array_equal(ss, ix.sketch_size); np.testing.assert_array_equal(gs, ix.genome_size)
    nq, cap = 600, 255
    qs = make_queries(nq)
    ptrs, lens = L.seq_arrays(qs)
    qset = C.c_void_p()
    L.check(lib.mk_qset_upload(ix._h, ptrs, lens, nq, C.byref(qset)))
    try:
        rows = plain_rows(ix, qset, nq, cap)
        got = mkd.gather_compact(rows)                           # dist.gather of a device tensor over RCCL
        torch.cuda.current_stream().synchronize()
        assert got.shape == (1, nq * (cap + 1)) and torch.equal(got[0], rows)
        hits, nh = mkd.merge_compact_on_device(ix, got.contiguous(), nq, cap, 10)
        L.check(lib.mk_sync(ix._h))
        want, _ = ix.query(qs, 10, 4, 1.0)
        nh = nh.cpu().numpy().view(np.uint32)
        assert (nh <= 10).all()                                  # no row overflowed its 255 slots
        hits = hits.cpu().numpy().view(mkd.HIT_DTYPE).reshape(nq, 10)
        for q in range(nq):
            assert [(int(x["genome"]), int(x["matches"])) for x in hits[q, :nh[q]]] == [(w.genome, w.matches) for w in want[q]], q
    finally:
        lib.mk_qset_free(ix._h, qset)


def test_native_communicator_of_one(pg, index):
    """The C ABI's own RCCL call sites at world size 1: communicator from a unique id, Bloom MIN all-reduce in place,
    size all-gather (id base 0, total G), ncclGather of the exchange rows, and the overlapped exchange of
    mk_qset_run_compact_gather in both of its forms (one ncclGather for a small set, four grouped send / recv blocks
    for >= 4096 queries) -- each equal to what the plain calls give."""
    import torch
    from miekki_amd import lib as L
    ix, lib = index, index._lib
    ident = (C.c_uint8 * 128)()
    L.check(lib.mk_comm_unique_id(ident))
    comm = C.c_void_p()
    L.check(lib.mk_comm_create(ix._h, 0, 1, ident, C.byref(comm)))
    try:
        assert lib.mk_comm_rank(comm) == 0 and lib.mk_comm_world(comm) == 1
        reach = lib.mk_bloom_reachable_bytes(ix._h)
        before = np.empty(reach, np.uint8)
        L.check(lib.mk_index_export_bloom(ix._h, 0, reach, before.ctypes.data))
        L.check(lib.mk_comm_sync_bloom(comm))
        after = np.empty(reach, np.uint8)
        L.check(lib.mk_index_export_bloom(ix._h, 0, reach, after.ctypes.data))
        np.testing.assert_array_equal(before, after)
        base, total = C.c_uint32(7), C.c_uint32(7)
        L.check(lib.mk_comm_share_sizes(comm, C.byref(base), C.byref(total)))
        assert (base.value, total.value) == (0, G)
        L.check(lib.mk_comm_barrier(comm))
        t = torch.tensor([1.5, -2.0], dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        L.check(lib.mk_comm_allreduce_max_f64(comm, t.data_ptr(), 2))
        L.check(lib.mk_sync(ix._h))
        assert t.tolist() == [1.5, -2.0]
        for nq, cap in ((300, 255), (5000, 255), (4097, 200)):
            qs = make_queries(nq)
            ptrs, lens = L.seq_arrays(qs)
            qset = C.c_void_p()
            L.check(lib.mk_qset_upload(ix._h, ptrs, lens, nq, C.byref(qset)))
            try:
                rows = plain_rows(ix, qset, nq, cap)
                recv = torch.full((nq * (cap + 1),), -1, dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                L.check(lib.mk_comm_gather_rows(comm, rows.data_ptr(), nq * (cap + 1), recv.data_ptr(), 0))
                L.check(lib.mk_sync(ix._h))
                assert torch.equal(recv, rows)
                rows2 = torch.zeros_like(rows)
                recv2 = torch.full_like(recv, -1)
                hits = torch.zeros((nq, 240), dtype=torch.uint8, device="cuda")
                nh = torch.zeros(nq, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                L.check(lib.mk_qset_invalidate(ix._h, qset))
                L.check(lib.mk_qset_run_compact_gather(ix._h, comm, qset, 10, 4, 1.0, cap, rows2.data_ptr(), recv2.data_ptr(), 0))
                # queued right behind, no wait in between: the context's stream is ordered behind the exchange
                L.check(lib.mk_merge_compact(ix._h, recv2.data_ptr(), 1, nq, cap, 10, hits.data_ptr(), nh.data_ptr()))
                L.check(lib.mk_sync(ix._h))
                assert torch.equal(rows2, rows) and torch.equal(recv2, rows)
                want, _ = ix.query(qs[:400], 10, 4, 1.0)
                nhh = nh.cpu().numpy().view(np.uint32)
                assert (nhh <= 10).all()
                hh = hits.cpu().numpy().view(np.dtype([("genome", "<u4"), ("matches", "<u4"), ("jaccard", "<f8"), ("intersection", "<f8")])).reshape(nq, 10)
                for q in range(min(nq, 400)):
                    assert [(int(x["genome"]), int(x["matches"])) for x in hh[q, :nhh[q]]] == [(w.genome, w.matches) for w in want[q]], q
            finally:
                lib.mk_qset_free(ix._h, qset)
    finally:
        lib.mk_comm_destroy(comm)
    assert lib.mk_set_genome_id_base(ix._h, 0) == 0


def run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("exchange", ["rccl", "torch"])
def test_bench_forced_collective_path_equals_the_plain_line(exchange):
    """bench.py --gpus 1 --force-collective: the `world > 1` code of the bench (RCCL process group, Bloom all-reduce, size
    all-gather, the gather inside the timed step -- the library's own, or torch's with its stream hand-over) at world
    size 1, against the plain one-GPU line on the same problem: same work, same checks, a rate within noise."""
    common = ["--genomes", "3000", "--queries", "20000", "--h", "20", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    plain = run_bench("--gpus", "1", *common)
    forced = run_bench("--gpus", "1", "--force-collective", "--exchange", exchange, *common)
    assert forced["merge"]["forced_collective_path"] and forced["merge"]["backend"] == "nccl" and forced["merge"]["ranks"] == 1
    assert ("mk_qset_run_compact_gather" in forced["merge"]["collective"]) == (exchange == "rccl")
    for r in (plain, forced):
        assert r["check"]["top_hit_is_source_genome_of_all_queries"] == r["check"]["queries"] == 20000
        assert r["check"]["device_heap_equals_host_heap_of_strided_sample"] == r["check"]["strided_sample"] == 2000
        assert r["merge"]["overflowed_queries"] == 0
    assert forced["config"]["active_partitions_per_query"] == plain["config"]["active_partitions_per_query"]
    assert forced["merge"]["gather_bytes_per_rank"] == 20000 * (forced["merge"]["cap"] + 1) * 8
    assert forced["value"] > 0.85 * plain["value"], (forced["value"], plain["value"])
