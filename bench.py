#!/usr/bin/env python3
"""bench.py -- query x genome fingerprint comparisons/s on MI355X.

One "step" = one complete pass of the hot path (query sketch + Bloom gate +
fingerprint scan + top-hit selection, then -- for N > 1 -- the single RCCL gather of
the per-query heap entrants in their 8-byte exchange form, and filter_results' heap on
rank 0) over a batch of synthetic 1 kb queries that is already resident in HBM, against
an index of synthetic 5 Mb genomes that was built on the device by the sketch kernels
before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--genomes G_TOTAL | --weak --genomes-per-gpu G] [--queries Q]
                    [--h H] [--fp-bits 8|16] [--rehearse]

N > 1: one rank per GPU under torch.distributed.run.  Invoked plainly
(`python bench.py --gpus 8`) the parent starts that launcher itself, before anything
in it touches the GPU, and relays rank 0's JSON line.  Default = BASELINE config 3 as
written: 100,000 genomes TOTAL, rank r owns the contiguous id range
shard_range(G, r, N) (strong scaling: N = 1 and N = 8 solve the same problem); `--weak`
keeps the genomes per GPU fixed instead.  Every rank scans all queries against its
shard; the entrant rows are gathered on rank 0.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
GENOME_LEN = 5_000_000
QUERY_LEN = 1000


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=100_000,
                    help="genomes in TOTAL, sharded over the GPUs (BASELINE config 3: 100,000)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --genomes-per-gpu on every GPU")
    ap.add_argument("--genomes-per-gpu", type=int, default=None,
                    help="implies --weak: this many genomes on every GPU (total = N x this)")
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--h", type=int, default=20)
    ap.add_argument("--fp-bits", type=int, default=8)
    ap.add_argument("--cap", type=int, default=0,
                    help="heap-entrant slots per query per rank (0 = 128 on one GPU, 96 when sharded)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: gloo collectives on host copies, every rank on device 0")
    ap.add_argument("--plan", action="store_true", help="print the launch plan / shard table as JSON and exit (no GPU)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: take the N > 1 code path anyway (a process group of one on the nccl backend, a "
                         "communicator of one): RCCL init, the Bloom all-reduce, the size all-gather and the entrant gather "
                         "all execute on the one GPU")
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="rccl: the library's own collectives (mk_comm_*: ncclGather & co. called from the C ABI, the gather "
                         "overlapped with the next chunk's scan); torch: torch.distributed collectives on device tensors")
    args = ap.parse_args(argv)
    if args.genomes_per_gpu is not None:
        args.weak = True
    if args.weak and args.genomes_per_gpu is None:
        args.genomes_per_gpu = args.genomes
    return args


def shard_table(args, world):
    """(first genome id, one past the last) per rank."""
    from miekki_amd.shard import shard_range
    if args.weak:
        return [(r * args.genomes_per_gpu, (r + 1) * args.genomes_per_gpu) for r in range(world)]
    return [shard_range(args.genomes, r, world) for r in range(world)]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, argv):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """Parent of a plain `bench.py --gpus N`: has not imported torch or touched the GPU; starts
    the ranks as children and relays their output (rank 0 prints the JSON line)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(launcher_command(args, argv), env=env)


def usable_cpus():
    """CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota (the GPU boxes hand a
    job 16 CPUs' worth of a 256-thread host: more threads than that only take turns on them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                    # cgroup v2
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:                                                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    usable = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return usable, n, quota


def cpu_baseline(h, host_cores):
    """The reference's own code timed on ALL of this host's cores (and on one, for the per-core figure),
    on bounded samples of the same workload, by oracle/_ref/ref_harness (the unmodified reference,
    compiled in the authoring container; kind "reference"):
      scan    query_sequences (Miekki.cpp:344-372) the way query_file's threads call it, 201 queries a
              batch, on an index padded to G_cpu genomes with a saturated Bloom filter;
      sketch  insert_sequences (Miekki.cpp:277-314) the way index_file_of_file's threads call it
              (546-581), eleven genomes a batch -- its append is the reference's global critical
              section, which is what bounds it on many cores.
    When that build is absent: the oracle's scalar restatement of the scan on the same sample shape,
    one core (kind "port"), and no sketch leg."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    usable, affinity, quota = usable_cpus()
    threads = max(1, usable)                          # every CPU this job may use (host_cores says what the machine has)
    # G_cpu: the regime the metric is quoted in (>= 10^4 genomes: the per-genome compare dominates the
    # O(2^h x batch) sweep); two batches of 201 queries per thread, ~10-20 s on the box's cores
    G_cpu = 20_000 if h >= 19 else 10_000
    # One batch per thread.  query_file's batches hold 201 queries; at -h 20 one such batch costs a thread ~5 s alone
    # and ~200 s when 256 threads run one each (measured, profiles/r3_cpu_baseline_scaling.txt: the O(2^h x batch)
    # strided sweep of Miekki.cpp:355-360 saturates the memory system), so the all-thread sample uses batches of
    # 32 -- the same work per query, a sample six times shorter.  The one-thread sample is a batch of 201.
    per, batch = (2, 201) if threads <= 64 else (1, 32 if h >= 19 else 201)
    shape = (f"synthetic 1 kb queries vs {G_cpu} genomes (4 sketched 5 Mb genomes, columns padded cyclically), "
             f"-h {h}, saturated Bloom")
    if os.path.exists(harness):
        try:
            # glibc serves the reference's 9 MB per-query vectors from its heaps instead of mmap (1.4x on 64 threads)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false", MALLOC_MMAP_THRESHOLD_="4294967296",
                       MALLOC_TRIM_THRESHOLD_="17179869184", MALLOC_TOP_PAD_="268435456")
            sys.stderr.write(f"[bench] cpu baseline: reference query_sequences on {threads} threads ...\n")
            out = subprocess.run([harness, "scanbench", str(h), str(G_cpu), str(per), str(threads), str(batch)],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env).stdout.decode()
            r = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            res = {"value": r["comparisons"] / r["seconds"], "unit": "comparisons/s", "cores": threads,
                   "host_cores": host_cores, "cpu_affinity": affinity, "cgroup_cpu_quota": quota, "kind": "reference",
                   "all_core": r["comparisons"] / r["seconds"],
                   "per_core": r["one_thread_comparisons"] / r["one_thread_seconds"],
                   "sample": f"reference query_sequences on {threads} threads, one batch of {r['batch']} queries each, "
                             f"{r['queries']} {shape}; per_core = one thread, one batch of 201 queries",
                   "sample_seconds": r["seconds"] + r["one_thread_seconds"]}
            try:
                # (the append is one critical section, ~0.2 s per genome at -h 20 however many threads: ~96 genomes = ~20 s)
                n_gen = max(16, min(4 * threads, 96))
                sys.stderr.write(f"[bench] cpu baseline: reference insert_sequences on {threads} threads ...\n")
                out = subprocess.run([harness, "sketchbench", str(h), str(n_gen), str(threads)],
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env).stdout.decode()
                k = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
                res["sketch"] = {"value": k["genomes"] / k["seconds"], "unit": "sketches/s", "cores": threads,
                                 "all_core": k["genomes"] / k["seconds"],
                                 "per_core": k["one_thread_genomes"] / k["one_thread_seconds"],
                                 "sample": f"reference insert_sequences on {threads} threads (batches of 11 as index_file_of_file "
                                           f"makes them), {k['genomes']} synthetic 5 Mb genomes from memory, -h {h}; "
                                           f"per_core = one thread, {k['one_thread_genomes']} genomes",
                                 "sample_seconds": k["seconds"] + k["one_thread_seconds"]}
            except Exception as e:
                sys.stderr.write(f"reference sketchbench unavailable: {e}\n")
            return res
        except Exception as e:                       # fall through to the port
            sys.stderr.write(f"reference scanbench unavailable: {e}\n")
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import scanbench
        nq1 = 201                                     # one core: one of the reference's batches
        G_port = min(G_cpu, 4_000)                    # the scalar port is slow: a smaller index, said so
        r = scanbench.run(h, G_port, nq1)
        v = r["comparisons"] / r["seconds"]
        return {"value": v, "unit": "comparisons/s", "cores": 1, "host_cores": host_cores, "kind": "port",
                "per_core": v, "all_core": None,
                "sample": f"oracle query_sequences (scalar C restatement, one core), {nq1} "
                          + shape.replace(str(G_cpu), str(G_port)),
                "sample_seconds": r["seconds"]}
    except Exception as e:
        sys.stderr.write(f"oracle scanbench unavailable: {e}\n")
        return None


def sketch_roofline(genomes, build_s, h, W):
    """The index build is bound by vector-instruction issue, not by memory (SURVEY 8d asks for k-mers/s and sketches/s with
    the HBM fraction as an informational number).  The issue ceiling comes from RECORDED evidence, and only when it belongs to
    this run: dynamic SQ_INSTS_VALU counts of the two big kernels (profiles/pmc_build.json, tools/pmc_build.py: one entry per
    (k, h, fingerprint width), stamped with the SHA-256 of build.hip) x the price of their instruction mix (profiles/isa_mix.json,
    tools/isa_mix.py: the kernels' opcodes from their ISA x per-opcode costs measured by tools/ubench -- NOT the kernel's own
    run time) / (1024 SIMDs x 2.4 GHz).  Per kernel: issue time over the kernel's time alone on the chip; for the batch: the sum
    of both over the measured time per 64-genome batch; beside it the same at the guide's peak of 2 cycles per wave-instruction."""
    import hashlib
    kmers = genomes * (GENOME_LEN - 31)
    out = {"bound": "valu-issue", "kmers_per_s": kmers / build_s, "sketches_per_s": genomes / build_s}
    # algorithmic HBM bytes per k-mer: 2-bit codes in, one item (4 bytes; 5 at 2-byte fingerprints) out and in again for the 15/16
    # of k-mers whose fingerprint is not "empty", W * 2^h fingerprint bytes per genome out, in and out again (transposer)
    item = 4 if W == 1 else 5
    bpk = 0.25 + 2 * item * 15 / 16 + 3.0 * W * (1 << h) / (GENOME_LEN - 31)
    out["hbm_bytes_per_kmer_algorithmic"] = bpk
    out["hbm_frac_informational"] = bpk * kmers / build_s / 1e9 / HBM_PEAK_GBS
    try:
        sha = hashlib.sha256(open(os.path.join(ROOT, "miekki_amd", "csrc", "build.hip"), "rb").read()).hexdigest()
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_build.json")))["entries"].get(f"k31_h{h}_fp{8 * W}")
        mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
        if pm is None:
            raise ValueError(f"no recorded counters for -k 31 -h {h}, {8 * W}-bit fingerprints")
        if pm["build_hip_sha256"] != sha:
            raise ValueError("the recorded counters belong to another generation of build.hip")
        if mix["build_hip_sha256"] != sha:
            raise ValueError("the recorded instruction mix belongs to another generation of build.hip")
        simds, clock = 256 * 4, 2.4e9
        per_batch_kmers = pm["batch_genomes"] * (pm["genome_len"] - 31)
        batch_s = build_s / max(genomes / pm["batch_genomes"], 1e-9)
        per_kernel, issue_total, issue_peak_total = {}, 0.0, 0.0
        for name, kd in pm["kernels"].items():
            mk = next((v for k, v in mix["kernels"].items() if k.startswith(name.replace(">", ","))), None)
            if mk is None or not kd.get("valu_wave_instructions_per_batch"):
                raise ValueError(f"no instruction mix for {name}")
            valu, cpi = kd["valu_wave_instructions_per_batch"], mk["weighted_cycles_per_valu"]
            issue_s, issue_peak_s = valu * cpi / (simds * clock), valu * 2.0 / (simds * clock)
            issue_total += issue_s; issue_peak_total += issue_peak_s
            per_kernel[name] = {"valu_instructions_per_kmer": valu * 64 / per_batch_kmers, "cycles_per_valu_from_isa_mix": cpi,
                                "issue_ms_per_batch": issue_s * 1e3, "alone_ms_per_batch": kd.get("alone_ms"),
                                "frac_alone": (issue_s * 1e3 / kd["alone_ms"]) if kd.get("alone_ms") else None,
                                "frac_alone_at_2_cycle_peak": (issue_peak_s * 1e3 / kd["alone_ms"]) if kd.get("alone_ms") else None}
        out.update({"per_kernel": per_kernel, "issue_ms_per_batch": issue_total * 1e3, "ms_per_batch": batch_s * 1e3,
                    "frac": issue_total / batch_s, "frac_at_2_cycle_peak": issue_peak_total / batch_s,
                    "derived_from": "recorded counters (profiles/pmc_build.json) and the kernels' ISA (profiles/isa_mix.json), both stamped with "
                                    "this build.hip; ms_per_batch is this run's",
                    "note": "frac = (SQ_INSTS_VALU of scatter + reduce per 64-genome batch x the cost of their instruction mix: opcode counts "
                            "from the ISA x per-opcode cycles measured by tools/ubench) / (1024 SIMDs x 2.4 GHz) over this run's time per batch; "
                            "frac_at_2_cycle_peak prices every vector instruction at the guide's 2 cycles; the two kernels run on two streams "
                            "and share the CUs zero-sum (DESIGN.md 4)",
                    "counters_source": pm["source"]})
    except Exception as e:
        out["frac"] = None
        out["frac_left_out_because"] = str(e)
    return out


def host_fed_build_rate(lib, L, device, h, fp_bits, packed, n=64, rounds=6, prefill=3072):
    """PCIe-inclusive build rate (DESIGN.md section 5 asks for it next to `value`, never as `value`): the same
    synthetic genomes, but handed over as HOST buffers (page-locked, mk_host_alloc) the way the `miekki`
    binary feeds them, into a scratch context of its own: packed (mk_index_append_packed: 2 bits per base, what
    the binary's readers produce) or as characters (mk_index_append).  Returns sketches/s over `rounds` batches."""
    import miekki_amd
    ix = miekki_amd.Miekki(31, h, fp_bits, 33, 200, device=device)
    try:
        ix.reserve(n * (rounds + 1) + prefill)
        buf = C.c_void_p()
        L.check(lib.mk_host_alloc(ix._h, n * GENOME_LEN, C.byref(buf)))
        L.check(lib.mk_probe_synth_genomes(ix._h, 10_000_000, n, GENOME_LEN, buf))
        pk = C.c_void_p()
        if packed:
            cw, xw = lib.mk_pack_code_words(GENOME_LEN), lib.mk_pack_except_words(GENOME_LEN)
            L.check(lib.mk_host_alloc(ix._h, n * (cw + xw) * 8, C.byref(pk)))
            arr = (L.PackedSeq * n)()
            for i in range(n):                                     # packed once, outside the timed region (the
                codes = pk.value + i * (cw + xw) * 8               # binary's reader threads do it while they parse)
                exc = codes + cw * 8
                if lib.mk_pack_append(codes, exc, 0, buf.value + i * GENOME_LEN, GENOME_LEN) != 0:
                    raise RuntimeError("synthetic genomes are plain ACGT")
                arr[i].codes, arr[i].except_, arr[i].len = codes, None, GENOME_LEN
                arr[i].head = C.string_at(buf.value + i * GENOME_LEN, 32)
            append = lambda: L.check(lib.mk_index_append_packed(ix._h, arr, n))
        else:
            ptrs = (C.c_char_p * n)(*[C.cast(buf.value + i * GENOME_LEN, C.c_char_p) for i in range(n)])
            lens = (C.c_uint64 * n)(*([GENOME_LEN] * n))
            append = lambda: L.check(lib.mk_index_append(ix._h, ptrs, lens, n))
        # steady state of a large build: a Bloom filter that has filled up (while it is young, the first ~2,000 genomes of
        # a collection, pass A works on every winner and the build kernels, not PCIe, are what such a run would time)
        ix.insert_synthetic(20_000_000, prefill, GENOME_LEN)
        append()                                                   # warm-up: scratch allocations
        L.check(lib.mk_sync(ix._h))
        t0 = time.perf_counter()
        for _ in range(rounds):
            append()
        L.check(lib.mk_sync(ix._h))
        dt = time.perf_counter() - t0
        lib.mk_host_free(ix._h, buf)
        if pk:
            lib.mk_host_free(ix._h, pk)
        return n * rounds / dt
    finally:
        ix.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.plan:
        world = args.gpus
        print(json.dumps({"launcher": launcher_command(args, [a for a in argv if a != "--plan"]) if world > 1 else None,
                          "scaling": "weak" if args.weak else "strong", "shards": shard_table(args, world)}))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    # stdout carries ONE JSON line and nothing else: RCCL prints a five-line version banner on stdout from the first
    # communicator of a process (torch's as well as the library's own), so descriptor 1 points at stderr until that line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libmiekki_hip has no CPU path")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_collective         # the N > 1 code path (forced: on a process group of one)
    if multi:
        if world == 1 and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = torch.device("cpu") if args.rehearse else torch.device("cuda", local_rank)
    native = multi and not args.rehearse and args.exchange == "rccl"   # (a rehearsal puts every rank on GPU 0: RCCL refuses that)

    import miekki_amd
    from miekki_amd import lib as L
    from miekki_amd import distributed as mkd
    lib = L.load_library()

    shards = shard_table(args, world)
    g0, g1 = shards[rank]
    G = g1 - g0
    G_total = shards[-1][1]
    Q = args.queries
    W = args.fp_bits // 8
    ix = miekki_amd.Miekki(31, args.h, args.fp_bits, 33, 200, device=local_rank, genome_id_base=g0)
    ix.reserve(G)
    t0 = time.time()
    done = 0
    while done < G:                                    # progress lines keep the runner's watchdog fed
        n = min(2048, G - done)
        ix.insert_synthetic(g0 + done, n, GENOME_LEN)
        done += n
        if rank == 0:
            sys.stderr.write(f"[bench] built {done}/{G} genomes in {time.time() - t0:.1f}s\n")
    L.check(lib.mk_sync(ix._h))
    build_s = time.time() - t0
    bst = ix.stats()
    build_s_max = build_s
    t_sync = time.time()
    comm = C.c_void_p()
    if multi:                                          # one global Bloom gate, as in a single-process build
        tb = torch.tensor([build_s], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        build_s_max = float(tb.item())
    if native:
        # the library's own communicator: rank 0 draws the id, the launcher's process group carries it
        ident = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            L.check(lib.mk_comm_unique_id(buf))
            ident = [bytes(buf)]
        dist.broadcast_object_list(ident, src=0)
        os.environ["MIEKKI_COMM_BANNER_TO_STDERR"] = "1"          # (stdout is one JSON line: RCCL's version banner to stderr)
        L.check(lib.mk_comm_create(ix._h, rank, world, ident[0], C.byref(comm)))
        L.check(lib.mk_comm_sync_bloom(comm))          # MIN all-reduce of rank-keyed cells (ncclAllReduce)
        base_, total_ = C.c_uint32(0), C.c_uint32(0)
        L.check(lib.mk_comm_share_sizes(comm, C.byref(base_), C.byref(total_)))      # ncclAllGather x 2
        assert (base_.value, total_.value) == (g0, shards[-1][1]), "shard table and exchanged sizes disagree"
        ss_all, gs_all = mkd.gather_sizes(ix.sketch_size, ix.genome_size, coll_dev)  # for the host-side check only
    else:
        if multi:
            mkd.sync_bloom(ix, device=coll_dev)
        ss_all, gs_all = mkd.share_sizes(ix, device=coll_dev)     # sizes of all genomes, once (for the merge)
    sync_s = time.time() - t_sync

    qs = C.c_void_p()
    L.check(lib.mk_qset_synthetic(ix._h, 0, Q, G_total, GENOME_LEN, QUERY_LEN, C.byref(qs)))
    from miekki_amd.shard import entrant_cap
    cap = args.cap or entrant_cap(10, max(b - a for a, b in shards))       # the largest shard decides the row width
    rw = cap + 1
    d_rows = torch.zeros(Q * rw, dtype=torch.int64, device="cuda")
    g_rows = None
    if multi and rank == 0:
        g_rows = torch.zeros((world, Q * rw), dtype=torch.int64, device=coll_dev)
    nres, min_score, min_inter = 10, 10, 100.0        # query_file's filter_results(.., 10, 10, 0.5*threshold)
    d_hits = torch.zeros((Q, nres * 24), dtype=torch.uint8, device="cuda")      # the step's product, on rank 0
    d_nhits = torch.zeros(Q, dtype=torch.int32, device="cuda")
    merge_s, gather_s = [0.0], [0.0]
    torch.cuda.synchronize()                          # torch's fills are on its own stream, the library has another

    def step():
        L.check(lib.mk_qset_invalidate(ix._h, qs))    # a step is a COMPLETE pass: sketch + gate are redone
        if native:
            # scan + selection + the one exchange step, all queued by the library: the rows of finished queries leave for
            # rank 0 (ncclGather / grouped ncclSend-ncclRecv blocks on the communicator's stream) while the next chunk
            # scans; the merge is queued right behind -- no host wait, no hand-over between torch's streams and ours
            L.check(lib.mk_qset_run_compact_gather(ix._h, comm, qs, nres, min_score, min_inter, cap, d_rows.data_ptr(),
                                                   g_rows.data_ptr() if rank == 0 else None, 0))
            if rank == 0:
                t_m = time.perf_counter()
                mkd.merge_compact_on_device(ix, g_rows, Q, cap, nres, out=(d_hits, d_nhits))
                L.check(lib.mk_sync(ix._h))
                merge_s[0] += time.perf_counter() - t_m
            else:
                L.check(lib.mk_sync(ix._h))
            return
        L.check(lib.mk_qset_run_compact(ix._h, qs, nres, min_score, min_inter, cap, d_rows.data_ptr()))
        L.check(lib.mk_sync(ix._h))
        rows = d_rows.view(1, -1)
        if multi:                                     # the one exchange step: heap entrants -> rank 0
            t_g = time.perf_counter()
            rows = mkd.gather_compact(d_rows.to(coll_dev), out=g_rows)
            if not args.rehearse:
                torch.cuda.current_stream().synchronize()
            gather_s[0] += time.perf_counter() - t_g
        if rank == 0:                                 # filter_results' heap over the rows in shard order (K6b)
            t_m = time.perf_counter()
            if args.rehearse and multi:
                rows = rows.cuda()
            torch.cuda.current_stream().synchronize() # the gather ran on torch's streams, the merge on the library's
            mkd.merge_compact_on_device(ix, rows.contiguous(), Q, cap, nres, out=(d_hits, d_nhits))
            L.check(lib.mk_sync(ix._h))
            merge_s[0] += time.perf_counter() - t_m

    for _ in range(args.warmup):
        step()
    ix.reset_stats()
    merge_s[0] = gather_s[0] = 0.0
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = ix.stats()
    # what every rank spent where (ms per step), so that a scaling record explains itself: max and min over ranks
    mine = [st["scan_ms"] / args.steps, st["filter_ms"] / args.steps, st["sketch_ms"] / args.steps,
            gather_s[0] / args.steps * 1e3, merge_s[0] / args.steps * 1e3, build_s]
    per_rank = [mine]
    if multi:
        tt = torch.tensor(mine, dtype=torch.float64, device=coll_dev)
        parts = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        per_rank = [[float(x) for x in p_.cpu()] for p_ in parts]
    rank_ms = {name: {"max": max(r[i] for r in per_rank), "min": min(r[i] for r in per_rank)}
               for i, name in enumerate(("scan", "select", "query_sketch", "gather", "merge"))}
    rank_ms["build_s"] = {"max": max(r[5] for r in per_rank), "min": min(r[5] for r in per_rank)}
    active = np.zeros(Q, np.uint32)
    L.check(lib.mk_qset_active(ix._h, qs, active.ctypes.data))
    a_sum = int(active.sum())
    comparisons_step_rank = a_sum * G
    comparisons_step = a_sum * G_total                # every rank has the same queries and the same (global) gate
    value = comparisons_step * args.steps / dt

    merged_ok = host_heap_ok = n_over = n_hit = top_all = None
    ns = 0
    if rank == 0:                                     # the step's hits: every query's top hit, the heap on a strided sample
        nh = d_nhits.cpu().numpy().view(np.uint32)
        n_over = int((nh == mkd.MERGE_OVERFLOW).sum())
        n_hit = int(((nh > 0) & (nh != mkd.MERGE_OVERFLOW)).sum())
        hits_all = d_hits[:Q].cpu().numpy().view(mkd.HIT_DTYPE).reshape(Q, nres)
        # query q was cut from genome q mod G_total: it must come out on top -- for ALL queries of the step
        good = (nh > 0) & (nh <= nres) & (hits_all[:, 0]["genome"] == (np.arange(Q, dtype=np.uint64) % G_total).astype(np.uint32))
        top_all = int(good.sum())
        # and the device heap must be the host's std:: heap over the same rows: 2,000 queries spread over the whole set
        # (every chunk of the slab schedule, every workgroup's share of the selection), not its first 2,000
        ns = min(Q, 2000)
        pick = (np.arange(ns, dtype=np.int64) * Q) // ns
        merged_ok = int(good[pick].sum())
        src = g_rows if multi else d_rows.view(1, -1)
        pick_d = torch.from_numpy(pick).to(src.device)
        rows_h = src.view(world, Q, rw)[:, pick_d].cpu().numpy().view(np.uint64).reshape(world, ns * rw)
        hits, overflow = mkd.merge_compact_host(rows_h, ns, cap, nres, ss_all, gs_all)
        host_heap_ok = sum(1 for i, q in enumerate(pick) if overflow[i] or
                           (nh[q] == len(hits[i]) and hits_all[q, :nh[q]].tobytes() == hits[i].tobytes()))

    if rank == 0:
        launches = max(1, int(st["scan_launches"]))
        algo_bytes_step = comparisons_step_rank * W + 4 * Q * G
        algo_per_launch = algo_bytes_step * args.steps / launches
        avg_launch_s = st["scan_ms"] / 1e3 / launches
        achieved = algo_per_launch / avg_launch_s / 1e9
        # the box's own read-only stream rate over the resident matrix, measured now
        gbps, nbytes = C.c_double(0), C.c_uint64(0)
        L.check(lib.mk_probe_stream_read(ix._h, 3, C.byref(gbps), C.byref(nbytes)))
        stream_gbs = gbps.value or None
        matrix_bytes = (1 << args.h) * G * W
        traffic = traffic_src = None
        try:                                           # PMC pass of this exact config, if one is committed
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            for e in pm["entries"]:
                if (e["genomes_per_gpu"], e["queries"], e["h"], e["fp_bits"]) == (G, Q, args.h, args.fp_bits):
                    # (per step in the file: a step's launches differ in size and the collective path cuts it its own way)
                    traffic = e["traffic_bytes_per_step"] * args.steps / launches
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this config; " \
                                  "L2<->fabric bytes, Infinity-Cache hits included)"
        except Exception:
            pass
        scaling = "weak" if args.weak else "strong"
        out = {
            "metric": "query x genome fingerprint comparisons/sec at -h %d" % args.h,
            "value": value, "unit": "comparisons/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8" if W == 1 else "u16", "data": "synthetic",
            "config": {"workload": "%d synthetic 5 Mb genomes in total, %s over %d GPU(s) (%d on rank 0), -k 31 -h %d, "
                                   "%d-bit fingerprints, %d x 1 kb queries scanned by every rank (%s scaling)"
                                   % (G_total, "sharded by contiguous id range" if world > 1 else "all", world, G,
                                      args.h, args.fp_bits, Q, scaling),
                       "genomes_per_gpu": G, "genomes_total": G_total, "queries": Q, "h": args.h,
                       "active_partitions_per_query": a_sum / max(Q, 1), "parallelism": "genome-shard x%d" % world},
            "roofline": {"bound": "hbm", "bound_detail": "infinity-cache / HBM read: the slab schedule serves repeat touches "
                                                         "of a row piece from the 256 MiB Infinity Cache",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "measured_stream_read_ceiling": stream_gbs,
                         "frac_of_measured_stream_read": (achieved / stream_gbs) if stream_gbs else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_floor_bytes": matrix_bytes,
                         "reuse_factor": algo_per_launch / matrix_bytes if matrix_bytes else None,
                         "served_by": "Infinity Cache + L2 (repeat touches of a row piece) over HBM (first touch)",
                         "hbm_side": {"bytes_per_launch_floor": matrix_bytes,
                                      "gbs_floor": matrix_bytes / avg_launch_s / 1e9,
                                      "frac_of_measured_stream_read": (matrix_bytes / avg_launch_s / 1e9 / stream_gbs) if stream_gbs else None,
                                      "note": "the least HBM itself must deliver: the shard's matrix once per launch; the true HBM share "
                                              "lies between this and `traffic`; no counter on this image isolates it: FETCH_SIZE counts "
                                              "Infinity-Cache hits, rocprofv3 -L lists no MALL / UMC byte counters, and TCC_EA0_RDREQ_DRAM_sum "
                                              "equals TCC_EA0_RDREQ_sum on this kernel (profiles/r3_pmc_scan_dram_counter.txt)"},
                         "kernel": "scan_slab_kernel" if st["scan_slab_launches"] else "scan_kernel",
                         "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                         "algo_bytes_per_launch": algo_per_launch,
                         "note": "achieved = algorithmic bytes (W per comparison + 4 per query x genome) / HIP-event launch time; "
                                 "hbm_floor_bytes = the shard's matrix once per launch, the least HBM can deliver; "
                                 "measured_stream_read_ceiling = mk_probe_stream_read over the resident matrix in this run"},
            "sketch": {"roofline": sketch_roofline(G, build_s, args.h, W),
                       "query_sketch_ms_per_step": st["sketch_ms"] / args.steps,
                       "index_build_s": build_s_max, "index_sketches_per_s": G_total / build_s_max,
                       "index_sketches_per_s_per_gpu": G / build_s,
                       "index_kmers_per_s": G_total * (GENOME_LEN - 31) / build_s_max,
                       "build_sketch_ms": bst["build_sketch_ms"], "build_finalize_ms": bst["build_finalize_ms"],
                       "bloom_and_sizes_sync_s": sync_s},
            "select": {"kernel": "select_kernel", "ms_per_step": st["filter_ms"] / args.steps},
            "merge": {"kernel": "merge_kernel", "ms_per_step": merge_s[0] / args.steps * 1e3, "overflowed_queries": n_over,
                      "cap": cap, "gather_bytes_per_rank": Q * rw * 8 if multi else 0,
                      "gather_ms_per_step": None if native else gather_s[0] / args.steps * 1e3,
                      "backend": (dist.get_backend() if multi else None),
                      "ranks": (dist.get_world_size() if multi else 1),
                      "forced_collective_path": bool(args.force_collective and world == 1),
                      "collective": (None if not multi else
                                     "mk_qset_run_compact_gather: ncclGather / grouped ncclSend-ncclRecv blocks from the C ABI (RCCL), queued "
                                     "on the communicator's stream beside the next chunk's scan; not timed apart (it hides behind the scan)"
                                     if native else "torch.distributed.gather over %s" % dist.get_backend()),
                      "per_rank_ms_per_step": rank_ms,
                      "note": "rank 0: filter_results heap over the (gathered) 8-byte entrant rows, inside the timed step"},
            "check": {"queries": Q, "queries_with_hits": n_hit, "top_hit_is_source_genome_of_all_queries": top_all,
                      "strided_sample": ns, "top_hit_is_source_genome_of_strided_sample": merged_ok,
                      "device_heap_equals_host_heap_of_strided_sample": host_heap_ok},
        }
        if world == 1:
            try:                                       # outside the timed region, a context of its own
                rate = host_fed_build_rate(lib, L, local_rank, args.h, args.fp_bits, packed=True, rounds=12)
                rate_c = host_fed_build_rate(lib, L, local_rank, args.h, args.fp_bits, packed=False, rounds=12)
                out["sketch"]["host_fed_sketches_per_s"] = rate
                out["sketch"]["host_fed_chars_sketches_per_s"] = rate_c
                out["sketch"]["host_fed_note"] = ("PCIe-inclusive: 12 batches of 64 x 5 Mb genomes from page-locked host buffers into an index of 3,072 "
                                                  "genomes (a Bloom filter that has filled up: the steady state of a large build), "
                                                  "packed 2 bits per base through mk_index_append_packed (%.1f GB/s over PCIe) and, "
                                                  "second figure, as characters through mk_index_append (%.1f GB/s); not part of `value`"
                                                  % (rate * GENOME_LEN / 4 / 1e9, rate_c * GENOME_LEN / 1e9))
            except Exception as e:
                sys.stderr.write(f"host-fed build sample skipped: {e}\n")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.h, os.cpu_count() or 1)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    lib.mk_qset_free(ix._h, qs)
    if comm:
        lib.mk_comm_destroy(comm)
    ix.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
