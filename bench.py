#!/usr/bin/env python3
"""bench.py -- query x genome fingerprint comparisons/s on MI355X.

One "step" = one pass of the hot path (query sketch + Bloom gate + fingerprint
scan + top-hit selection, then -- for N > 1 -- the single RCCL gather of the
per-query heap entrants) over a batch of synthetic 1 kb queries that is already
resident in HBM, against an index of synthetic 5 Mb genomes that was built on the
device by the sketch kernels before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--genomes-per-gpu G] [--queries Q] [--h H] [--fp-bits 8|16]

For N > 1 it is launched by torch.distributed.run, one rank per GPU; genomes are
sharded by rank (rank r owns ids [r*G, (r+1)*G)), every rank scans all queries
against its shard, and the candidate rows are gathered on rank 0 (weak scaling:
per-GPU work is fixed, total genomes = N * G).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
STREAM_READ_GBS = 6249.0     # read-only stream ceiling measured on the box (tools/scan_tune)
GENOME_LEN = 5_000_000
QUERY_LEN = 1000


def cpu_baseline(h, rank0_cores):
    """Reference scan timed on this host's cores on a bounded sample of the same
    workload (oracle/_ref/ref_harness scanbench = the real reference's
    query_sequences), or the oracle port when the reference build is absent."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    threads = max(1, min(rank0_cores, 16))
    # sized for roughly 10-15 s of scanning on 16 threads (the reference batches 201 queries per thread)
    G_cpu, nq = 10_000, 8 * 201 * threads
    if h >= 20:
        G_cpu, nq = 2_000, 4 * 201 * threads
    sample = f"reference query_sequences: {nq} synthetic 1 kb queries vs {G_cpu} genomes, -h {h}, saturated Bloom"
    if os.path.exists(harness):
        try:
            out = subprocess.run([harness, "scanbench", str(h), str(G_cpu), str(nq), str(threads)],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900).stdout.decode()
            line = [l for l in out.splitlines() if l.startswith("{")][-1]
            r = json.loads(line)
            return {"value": r["comparisons"] / r["seconds"], "unit": "comparisons/s", "cores": threads,
                    "kind": "reference", "sample": sample, "sample_seconds": r["seconds"]}
        except Exception as e:                       # fall through to the port
            sys.stderr.write(f"reference scanbench unavailable: {e}\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import synth
    from oracle import oracle as orc
    Gp, nqp, hp = 64, 32, min(h, 17)
    o = orc.OracleMiekki(31, hp, 8, 33, 200)
    o.insert_sequences([synth.genome_bases(g, 0, 1_000_000) for g in range(Gp)])
    qs = [synth.genome_bases(*synth.query_origin(q, Gp, 1_000_000, QUERY_LEN), QUERY_LEN) for q in range(nqp)]
    act = sum(o.query_sequence(s)[1] for s in qs)
    t0 = time.time()
    o.query_sequences(qs)
    dt = time.time() - t0
    return {"value": act * Gp / dt, "unit": "comparisons/s", "cores": 1, "kind": "port",
            "sample": f"oracle query_sequences: {nqp} queries vs {Gp} x 1 Mb genomes, -h {hp}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes-per-gpu", type=int, default=100_000,
                    help="default = BASELINE config 3 (100,000 genomes, 105 GB matrix) on every GPU")
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--h", type=int, default=20)
    ap.add_argument("--fp-bits", type=int, default=8)
    ap.add_argument("--cap", type=int, default=128, help="heap-entrant slots per query per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: gloo collectives on host copies, every rank on device 0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libmiekki_hip has no CPU path")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = torch.device("cpu") if args.rehearse else torch.device("cuda", local_rank)

    import miekki_amd
    from miekki_amd import lib as L
    lib = L.load_library()

    G = args.genomes_per_gpu
    G_total = G * world
    Q = args.queries
    W = args.fp_bits // 8
    ix = miekki_amd.Miekki(31, args.h, args.fp_bits, 33, 200, device=local_rank, genome_id_base=rank * G)
    ix.reserve(G)
    t0 = time.time()
    done = 0
    while done < G:                                    # progress lines keep the runner's watchdog fed
        n = min(2048, G - done)
        ix.insert_synthetic(rank * G + done, n, GENOME_LEN)
        done += n
        if rank == 0:
            sys.stderr.write(f"[bench] built {done}/{G} genomes in {time.time() - t0:.1f}s\n")
    L.check(lib.mk_sync(ix._h))
    build_s = time.time() - t0
    bst = ix.stats()
    build_s_max = build_s
    if world > 1:                                      # one global Bloom gate, as in a single-process build
        from miekki_amd import distributed as mkd
        tb = torch.tensor([build_s], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        build_s_max = float(tb.item())
        mkd.sync_bloom(ix, device=coll_dev)

    qs = C.c_void_p()
    L.check(lib.mk_qset_synthetic(ix._h, 0, Q, G_total, GENOME_LEN, QUERY_LEN, C.byref(qs)))
    cap = args.cap
    d_count = torch.zeros(Q, dtype=torch.int32, device="cuda")
    d_cand = torch.zeros(Q * cap * 24, dtype=torch.uint8, device="cuda")
    from miekki_amd import distributed as mkd
    g_rows = None
    if world > 1 and rank == 0:
        g_rows = (torch.zeros((world, Q), dtype=torch.int32, device=coll_dev),
                  torch.zeros((world, Q * cap * 24), dtype=torch.uint8, device=coll_dev))
    nres, min_score, min_inter = 10, 10, 100.0        # query_file's filter_results(.., 10, 10, 0.5*threshold)
    d_hits = torch.zeros((Q, nres * 24), dtype=torch.uint8, device="cuda")      # the step's product, on rank 0
    d_nhits = torch.zeros(Q, dtype=torch.int32, device="cuda")
    merge_s = [0.0]
    torch.cuda.synchronize()                          # torch's fills are on its own stream, the library has another

    def step():
        L.check(lib.mk_qset_run(ix._h, qs, nres, min_score, min_inter, cap, d_count.data_ptr(), d_cand.data_ptr()))
        L.check(lib.mk_sync(ix._h))
        rows_c, rows_d = d_count, d_cand
        if world > 1:                                 # the one exchange step: heap entrants -> rank 0
            rows_c, rows_d = mkd.gather_rows(d_count.to(coll_dev), d_cand.to(coll_dev), out=g_rows)
        if rank == 0:                                 # filter_results' heap over the rows in shard order (K6b)
            t_m = time.perf_counter()
            if args.rehearse and world > 1:
                rows_c, rows_d = rows_c.cuda(), rows_d.cuda()
            torch.cuda.current_stream().synchronize() # the gather ran on torch's streams, the merge on the library's
            L.check(lib.mk_merge_entrants(ix._h, rows_c.data_ptr(), rows_d.data_ptr(), world, Q, cap, nres,
                                          d_hits.data_ptr(), d_nhits.data_ptr()))
            L.check(lib.mk_sync(ix._h))
            merge_s[0] += time.perf_counter() - t_m

    for _ in range(args.warmup):
        step()
    ix.reset_stats()
    merge_s[0] = 0.0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    merged_ok = host_heap_ok = n_over = None
    if rank == 0:                                     # the step's hits, checked on a sample
        ns = min(Q, 2000)
        nh = d_nhits.cpu().numpy().view(np.uint32)
        n_over = int((nh == mkd.MERGE_OVERFLOW).sum())
        hits_dev = d_hits[:ns].cpu().numpy().view(mkd.HIT_DTYPE).reshape(ns, nres)
        # query q was cut from genome q mod G_total: it must come out on top
        merged_ok = sum(1 for q in range(ns) if 0 < nh[q] <= nres and int(hits_dev[q, 0]["genome"]) == q % G_total)
        # and the device heap must be the host's std:: heap over the same rows
        src_count = g_rows[0] if world > 1 else d_count.view(1, Q)
        src_cand = g_rows[1] if world > 1 else d_cand.view(1, -1)
        counts = src_count.cpu().numpy()[:, :ns]
        cands = src_cand.view(world, Q, cap * 24)[:, :ns].cpu().numpy()
        hits, overflow = mkd.merge_candidates(counts, cands.reshape(world, -1), cap, nres)
        host_heap_ok = sum(1 for q in range(ns) if overflow[q] or
                           (nh[q] == len(hits[q]) and hits_dev[q, :nh[q]].tobytes() == hits[q].tobytes()))
    st = ix.stats()
    active = np.zeros(Q, np.uint32)
    L.check(lib.mk_qset_active(ix._h, qs, active.ctypes.data))
    a_sum = int(active.sum())
    comparisons_step_rank = a_sum * G
    comparisons_step = comparisons_step_rank * world
    value = comparisons_step * args.steps / dt

    # sanity: every query's own genome must be among its candidates on the owning rank
    cnt = d_count.cpu().numpy()
    n_hit = int((cnt > 0).sum())

    if rank == 0:
        launches = max(1, int(st["scan_launches"]))
        algo_bytes_step = comparisons_step_rank * W + 4 * Q * G
        algo_per_launch = algo_bytes_step * args.steps / launches
        avg_launch_s = st["scan_ms"] / 1e3 / launches
        achieved = algo_per_launch / avg_launch_s / 1e9
        traffic = None
        try:                                           # PMC pass of this exact config, if one is committed
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            for e in pm["entries"]:
                if (e["genomes_per_gpu"], e["queries"], e["h"], e["fp_bits"]) == (G, Q, args.h, args.fp_bits):
                    traffic = e["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "query x genome fingerprint comparisons/sec at -h %d" % args.h,
            "value": value, "unit": "comparisons/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8" if W == 1 else "u16", "data": "synthetic",
            "config": {"workload": "%d synthetic 5 Mb genomes per GPU (%d total), -k 31 -h %d, %d-bit fingerprints, "
                                   "%d x 1 kb queries scanned by every rank" % (G, G_total, args.h, args.fp_bits, Q),
                       "genomes_per_gpu": G, "genomes_total": G_total, "queries": Q, "h": args.h,
                       "active_partitions_per_query": a_sum / max(Q, 1), "parallelism": "genome-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "measured_stream_read_ceiling": STREAM_READ_GBS,
                         "kernel": "scan_slab_kernel" if st["scan_slab_launches"] else "scan_kernel", "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                         "algo_bytes_per_launch": algo_per_launch},
            "sketch": {"query_sketch_ms_per_step": st["sketch_ms"] / args.steps,
                       "index_build_s": build_s_max, "index_sketches_per_s": G_total / build_s_max,
                       "index_sketches_per_s_per_gpu": G / build_s,
                       "index_kmers_per_s": world * bst["build_kmers"] / build_s_max,
                       "build_sketch_ms": bst["build_sketch_ms"], "build_finalize_ms": bst["build_finalize_ms"]},
            "select": {"kernel": "select_kernel", "ms_per_step": st["filter_ms"] / args.steps},
            "merge": {"kernel": "merge_kernel", "ms_per_step": merge_s[0] / args.steps * 1e3, "overflowed_queries": n_over,
                      "note": "rank 0: stream hand-over + filter_results heap over the (gathered) entrant rows, inside the timed step"},
            "check": {"queries_with_candidates_on_rank0": n_hit, "top_hit_is_source_genome_of_first_2000": merged_ok,
                      "device_heap_equals_host_heap_of_first_2000": host_heap_ok},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.h, os.cpu_count() or 1)
        print(json.dumps(out))
    lib.mk_qset_free(ix._h, qs)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
