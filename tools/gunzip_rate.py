#!/usr/bin/env python3
"""Rate of the device's gzip inflater (mk_gz_inflate) on gzip'd synthetic 5 Mb FASTA files.
    python tools/gunzip_rate.py [streams] [distinct] [level]
Wall time includes the upload of the files and the download of the text; run under `rocprofv3 --kernel-trace --stats` for
the two kernels' own times."""
import gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
import miekki_amd
from miekki_amd import lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
Lg = 5_000_000
texts = [synth.fasta(f"genome{g}", synth.genome_bases(g, 0, Lg)) for g in range(D)]
blobs = [gzip.compress(t, level) for t in texts]
ix = miekki_amd.Miekki(31, 14, 8, 33, 200)
batch = [blobs[i % D] for i in range(N)]
rooms = [len(texts[i % D]) for i in range(N)]
for rep in range(2):
    t0 = time.time()
    got, status = L.gz_inflate(ix._h, batch, rooms)
    dt = time.time() - t0
    if not os.environ.get("MIEKKI_NOCHECK"):                 # (timing experiments with kernels that leave work out)
        assert all(s == 0 for s in status), status[:8]
        assert all(got[i] == texts[i % D] for i in range(0, N, max(1, N // 16)))
    print(f"run {rep}: {N} streams ({sum(len(b) for b in batch) / 1e6:.0f} MB gz -> {sum(rooms) / 1e6:.0f} MB text) in {dt:.3f} s = {N / dt:.0f} files/s incl. copies", flush=True)
ix.close()
