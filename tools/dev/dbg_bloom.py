import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import synth
import miekki_amd as hip
from miekki_amd import lib as L
from oracle import oracle as orc
k, h = 23, 7
lens = [20000, 5000, k, 60000, k, 300]
seqs = [synth.genome_bases(5100 + g, 0, n) for g, n in enumerate(lens)]
for sub in ([0, 1, 2, 3, 4, 5], [5], [0, 5], [3, 5]):
    ss = [seqs[i] for i in sub]
    o = orc.OracleMiekki(k, h, 8, 32, 0); o.insert_sequences(ss)
    ix = hip.Miekki(k, h, 8, 32, 0); ix.insert_sequences(ss)
    nb = ix.bloom_size // 8
    b = np.zeros(nb, np.uint8)
    L.check(ix._lib.mk_index_export_bloom(ix._h, 0, nb, b.ctypes.data))
    ob = np.array(o.bloom[:nb])
    d = np.nonzero(b != ob)[0]
    print(sub, "cells set: hip", int((b != 0).sum()), "oracle", int((ob != 0).sum()), "differing", len(d), d[:10], b[d[:10]], ob[d[:10]])
    ix.close()
o = orc.OracleMiekki(k, h, 8, 32, 0); o.insert_sequences(seqs)
ix = hip.Miekki(k, h, 8, 32, 0); ix.insert_sequences(seqs)
a = b"".join(ix.serialize()); w = o.serialize().tobytes()
print("index stream equal:", a == w, len(a), len(w))
qs = [seqs[3][:900], seqs[3][:k], seqs[0][:6000], seqs[5]]
got = ix.query_sequences(qs); want = o.query_sequences(qs)
print("scores equal:", np.array_equal(got, want)); print(got); print(want)
hits, _ = ix.query(qs, 1, 0, 25.0)
print([[(x.genome, x.matches) for x in hq] for hq in hits])
hits, _ = ix.query([seqs[5]], 1, 0, 25.0)
print("alone:", [[(x.genome, x.matches) for x in hq] for hq in hits])
