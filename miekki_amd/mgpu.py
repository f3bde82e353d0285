"""Multi-GPU `miekki -l … (-a | -A) … [-e] -o …`: genome-sharded index, one gather of top hits.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m miekki_amd.mgpu -l genomes.txt -a queries.fa -o out.txt -h 20

One process per GPU (backend "nccl" = RCCL over xGMI).  Rank r indexes the r-th
contiguous slice of the genome list on its own GPU -- genome ids are those of a
single-process run because the slices are contiguous and every rank learns how many
genomes the ranks before it kept (one tiny all-gather at build time).  The Bloom
filters are merged byte-exactly (distributed.sync_bloom), every rank scans all
queries against its shard, and the per-query heap entrants are gathered on rank 0 as 8-byte (genome, matches) records,
which replays the reference heap and writes the reference's out.txt
(Miekki.cpp:440-444).  `--rehearse` runs the same program with gloo collectives and
every rank on GPU 0 (single-GPU boxes, tests).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import distributed as mkd
from . import lib as L
from .index import Miekki, SimilarityScore, _read_text


def main(argv=None):
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("-l", required=True); ap.add_argument("-a"); ap.add_argument("-A")
    ap.add_argument("-o", default="out.txt")
    ap.add_argument("-h", type=int, default=17); ap.add_argument("-k", type=int, default=31)
    ap.add_argument("-f", type=int, default=3); ap.add_argument("-b", type=int, default=33)
    ap.add_argument("-s", type=float, default=200.0); ap.add_argument("-t", type=int, default=8)
    ap.add_argument("-e", action="store_true", help="exact mode: real intersection computed on hits (Miekki.cpp:723-759)")
    ap.add_argument("--cap", type=int, default=96)
    ap.add_argument("--rehearse", action="store_true")
    ap.add_argument("--help", action="help")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    coll = torch.device("cpu") if args.rehearse else torch.device("cuda", local)
    lib = L.load_library()

    # ---- build: contiguous slice of the list per rank (Miekki.cpp:540-588 semantics per file)
    files = [f.decode() for f in _read_text(args.l).split(b"\n") if len(f) > 3]
    f0, f1 = mkd.shard_range(len(files), rank, world)
    seqs, my_files = [], []
    for fn in files[f0:f1]:
        if not os.path.exists(fn):
            print(f"Missed file: {fn}", flush=True)
            continue
        ref = b"".join(l for l in _read_text(fn).split(b"\n") if not l.startswith(b">"))
        if len(ref) >= args.k:
            seqs.append(ref)
            my_files.append(fn)
    kept = torch.tensor([len(seqs)], dtype=torch.int64, device=coll)
    if world > 1:
        all_kept = [torch.zeros_like(kept) for _ in range(world)]
        dist.all_gather(all_kept, kept)
        base = int(sum(int(t.item()) for t in all_kept[:rank]))
        total = int(sum(int(t.item()) for t in all_kept))
    else:
        base, total = 0, len(seqs)
    ix = Miekki(args.k, args.h, 5 + args.f, args.b, int(args.s), device=local, genome_id_base=base)
    ix.reserve(max(len(seqs), 1))                                # no doubling re-layouts (each holds old and new matrix at once)
    for i in range(0, len(seqs), 64):
        ix.insert_sequences(seqs[i:i + 64])
    del seqs
    if world > 1:
        mkd.sync_bloom(ix, device=coll)
    mkd.share_sizes(ix, device=coll)                               # sizes of all genomes, once: the merge recomputes
    if rank == 0:                                                  # jaccard / intersection from (genome, matches) records
        print(f"Reference indexed: {total}", flush=True)

    # ---- query: every rank scans all records against its shard (Miekki.cpp:426-483)
    if args.a:
        lines = _read_text(args.a).split(b"\n")
        recs = [(lines[i], lines[i + 1] if i + 1 < len(lines) else b"") for i in range(0, len(lines), 2)]
        recs = [(h, s) for h, s in recs if len(s) >= args.k]
    elif args.A:                                                 # whole files as queries (Miekki.cpp:592-612, 487-514)
        recs = []
        for fn in [f.decode() for f in _read_text(args.A).split(b"\n") if len(f) > 3]:
            if not os.path.exists(fn):
                if rank == 0:
                    print("File problem", flush=True)
                continue
            ref = b"".join(l for l in _read_text(fn).split(b"\n") if not l.startswith(b">"))
            if len(ref) >= args.k:
                recs.append((fn.encode(), ref))
    else:
        raise SystemExit("No query file, No queries")
    nres, min_score, min_inter, cap = 10, 10, 0.5 * int(args.s), args.cap
    if args.e and args.A:
        raise SystemExit("-A -e is not implemented in the multi-GPU driver")
    if args.e:                                                   # query_file_exact: filter_results(.., 5, 10, threshold)
        recs = [(h, s) for h, s in recs if s[:1] in (b"A", b"C", b"G", b"T", b"N")]
        nres, min_inter = 5, float(int(args.s))
    out = open(args.o, "wb") if rank == 0 else None
    step = 64 if args.A else 16384
    for b0 in range(0, len(recs), step):
        chunk = recs[b0:b0 + step]
        nq = len(chunk)
        rw = cap + 1                                             # 64-bit words of one exchange row
        d_rows = torch.zeros(nq * rw, dtype=torch.int64, device="cuda")
        # short records (at most 4,096 k-mers) and long ones are run as two sets, so that the
        # short ones keep the slab schedule (mk_query does the same for its batches)
        short = [i for i, (_, s) in enumerate(chunk) if len(s) <= args.k + 4096]
        long_ = [i for i, (_, s) in enumerate(chunk) if len(s) > args.k + 4096]
        for part in (short, long_):
            if not part:
                continue
            whole = len(part) == nq
            ptrs, lens = L.seq_arrays([chunk[i][1] for i in part])
            qs = C.c_void_p()
            L.check(lib.mk_qset_upload(ix._h, ptrs, lens, len(part), C.byref(qs)))
            p_rows = d_rows if whole else torch.zeros(len(part) * rw, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            L.check(lib.mk_qset_run_compact(ix._h, qs, nres, min_score, float(min_inter), cap, p_rows.data_ptr()))
            L.check(lib.mk_sync(ix._h))
            lib.mk_qset_free(ix._h, qs)
            if not whole:
                idx = torch.tensor(part, dtype=torch.int64, device="cuda")
                d_rows.view(nq, rw)[idx] = p_rows.view(len(part), rw)
                torch.cuda.synchronize()
        if world > 1:                                            # the one exchange step: 8-byte entrant records
            rows = mkd.gather_compact(d_rows.to(coll))
        else:
            rows = d_rows.view(1, -1)
        # rank 0: filter_results' heap over the rows in shard order, on the GPU (K6b).
        # Rows that overflowed on some shard: every rank answers them again with its
        # complete (not just entrant) candidate list, gathered as objects
        over = None
        if rank == 0:
            if not rows.is_cuda:
                rows = rows.cuda()
            torch.cuda.current_stream().synchronize()
            d_hits, d_nh = mkd.merge_compact_on_device(ix, rows.contiguous(), nq, cap, nres)
            L.check(lib.mk_sync(ix._h))
            nh = d_nh.cpu().numpy().view(np.uint32)
            hits = d_hits.cpu().numpy().view(mkd.HIT_DTYPE).reshape(nq, max(nres, 1))
            over = np.flatnonzero(nh == mkd.MERGE_OVERFLOW).tolist()
        if world > 1:
            box = [over]
            dist.broadcast_object_list(box, src=0)
            over = box[0]
        full = {}
        if over:
            scores = ix.query_sequences([chunk[q][1] for q in over])
            ss, gs = ix.sketch_size.astype(np.float64), ix.genome_size.astype(np.float64)
            mine = {}
            for j, q in enumerate(over):
                with np.errstate(divide="ignore", invalid="ignore"):
                    jac = scores[j] / ss
                    inter = jac * gs
                keep = np.flatnonzero((scores[j] >= min_score) & ~(inter < min_inter))
                mine[q] = [(int(g) + base, int(scores[j][g]), float(jac[g]), float(inter[g])) for g in keep]
            gathered = [None] * world
            if world > 1:
                dist.gather_object(mine, gathered if rank == 0 else None, dst=0)
            else:
                gathered = [mine]
            if rank == 0:
                for q in over:
                    full[q] = [c for part in gathered for c in part[q]]
        if rank == 0:
            text = []
            for q, (head, _) in enumerate(chunk):
                if q in full:
                    buf = (L.Hit * max(len(full[q]), 1))(*[L.Hit(*c) for c in full[q]])
                    res = (L.Hit * nres)()
                    n = lib.mk_filter_candidates(buf, len(full[q]), nres, res)
                    row = [SimilarityScore(res[i].genome, res[i].matches, res[i].jaccard, res[i].intersection) for i in range(n)]
                else:
                    row = [SimilarityScore(int(h["genome"]), int(h["matches"]), float(h["jaccard"]), float(h["intersection"]))
                           for h in hits[q, :nh[q]]]
                if args.e:
                    text.append(row)
                elif args.A:                                     # a line only when there are hits (Miekki.cpp:506-511)
                    text.append(Miekki.format_hits(head, row) if row else b"")
                else:
                    text.append(Miekki.format_hits(head, row))
            if not args.e:
                out.write(b"".join(text))
        if args.e:
            # K7 on the rank that owns the genome: rank 0 hands every owner its (genome, queries)
            # work list, the owners answer with (inter, union) per query, rank 0 prints
            # ground_truth_batch's lines (Miekki.cpp:843-855)
            work = [dict() for _ in range(world)]
            if rank == 0:
                bases = [0]
                for t in (all_kept if world > 1 else [kept]):
                    bases.append(bases[-1] + int(t.item()))
                for q, row in enumerate(text):
                    for h in row:
                        owner = max(r for r in range(world) if bases[r] <= h.genome)
                        work[owner].setdefault(h.genome - bases[owner], []).append((q, h.jaccard, h.intersection))
            if world > 1:
                mine = [None]
                dist.scatter_object_list(mine, work if rank == 0 else None, src=0)
                mine = mine[0]
            else:
                mine = work[0]
            answers = []
            for g_local, items in sorted(mine.items()):
                inter, uni = ix.ground_truth_batch([chunk[q][1] for q, _, _ in items], _read_text(my_files[g_local]))
                for (q, jac, est), ni, nu in zip(items, inter, uni):
                    if ni > 0:
                        answers.append((q, int(ni) / int(nu), jac, int(ni), est, my_files[g_local]))
            gathered = [None] * world
            if world > 1:
                dist.gather_object(answers, gathered if rank == 0 else None, dst=0)
            else:
                gathered = [answers]
            if rank == 0:
                for part in gathered:
                    for q, real, jac, ni, est, fn in part:
                        out.write(("%g\t%g\t%g\t%g\t%s\t%s\n" % (real, jac, ni, est, chunk[q][0].decode(), fn)).encode())
    if out:
        out.close()
    ix.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
