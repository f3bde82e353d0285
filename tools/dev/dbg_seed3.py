import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, miekki_amd
from oracle import oracle as orc

seed = 3
rng = np.random.default_rng(31000 + seed)
k = int(rng.integers(9, 32)); h = int(rng.integers(6, 15)); fpb = int(rng.choice([8, 16])); thr = int(rng.integers(0, 30))
print("k h fpb", k, h, fpb)
o = orc.OracleMiekki(k, h, fpb, 32, thr)
ix = miekki_amd.Miekki(k, h, fpb, 32, thr)
pool = []; kinds = []
next_id = 0
def check_queries():
    if not pool: return
    for _ in range(int(rng.integers(1, 6))):
        src = pool[int(rng.integers(0, len(pool)))]
        n = int(rng.choice([k, 200, 1500, 4096 + k + 300, len(src)]))
        off = int(rng.integers(0, max(1, len(src) - min(n, len(src)) + 1)))
    int(rng.choice([1, 5, 10])); int(rng.integers(1, 4)); float(rng.choice([0.0, 10.0]))
for step in range(int(rng.integers(5, 10))):
    op = rng.choice(["append", "append", "synthetic", "query", "reload"])
    print("step", step, op)
    if op == "append":
        n = int(rng.choice([0, 1, 3, 20, 64, 70]))
        seqs = [synth.genome_bases(40_000 + 500 * seed + next_id + i, 0, int(rng.choice([k, k + 5, 800, 6000, 25_000]))) for i in range(n)]
        next_id += n
        cut = int(rng.integers(0, n + 1))
        ix.insert_sequences(seqs[:cut]); ix.insert_sequences(seqs[cut:])
        o.insert_sequences(seqs)
        pool += seqs; kinds += [("chars", step, i, cut)for i in range(n)]
    elif op == "synthetic":
        n, length = int(rng.integers(1, 9)), int(rng.choice([k + 1, 3000, 12_000]))
        first = 900_000 + 100 * seed + next_id
        ix.insert_synthetic(first, n, length)
        seqs = [synth.genome_bases(first + i, 0, length) for i in range(n)]
        o.insert_sequences(seqs)
        pool += seqs; next_id += n; kinds += [("synth", step, i, first + i) for i in range(n)]
    elif op == "query":
        check_queries()
    else:
        pass   # reload skipped: serialize + load (not needed to find the first mismatch)
    ss, oss = ix.sketch_size, o.sketch_size
    bad = np.flatnonzero(ss != oss)
    if bad.size:
        print("MISMATCH after step", step, "genomes", bad, [(kinds[g], len(pool[g]), int(ss[g]), int(oss[g])) for g in bad])
        break
G = ix.index_size
P = 1 << h
cols = np.frombuffer(b"".join(ix.serialize()), np.uint8, P * G * (fpb // 8), 39).reshape(P, G * (fpb // 8))
ocols = o.columns()
for g in np.flatnonzero(ix.sketch_size != o.sketch_size):
    d = np.flatnonzero(cols[:, g] != ocols[:, g])
    print("genome", g, "differs in partitions", d[:40], "ours", cols[d[:40], g], "oracle", ocols[d[:40], g])
    fp, hs, act = o.minhash_sketch_partition(pool[g])
    # positions of the oracle's winners for the differing partitions
    seq = pool[g]
    print("  bins of differing partitions:", (d[:40] >> 12))
    # the same genome alone, three ways
    for how in ("synthetic", "chars", "packed"):
        a = miekki_amd.Miekki(k, h, fpb, 32, thr)
        if how == "synthetic" and kinds[g][0] == "synth": a.insert_synthetic(kinds[g][3], 1, len(seq))
        elif how == "chars": a.insert_sequences([seq])
        elif how == "packed": a.insert_sequences_packed([seq])
        else: a.close(); continue
        print("   alone via", how, int(a.sketch_size[0]), "oracle", act)
        a.close()
ix.close()
