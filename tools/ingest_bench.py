#!/usr/bin/env python3
"""Steady-state rate of `miekki -l` (host parse + H2D + device build) on many FASTA files.
    python tools/ingest_bench.py [n_genomes] [threads] [distinct] [gz]
The list names n_genomes files; only `distinct` different ones are written, the rest are
symlinks to them (the page cache is warm either way; the device work per genome is the same while the Bloom filter is
young -- it stays young for ever when the same 64 genomes come again and again, and fills up within ~2,000 distinct ones:
`distinct` = n_genomes is the collection a user has).
gz: the files are gzip members (level 6, what NCBI ships and what the reference's zstr reader inflates on the fly,
zstr.hpp:78-82): every reader thread then spends its time in zlib's inflate."""
import gzip, os, re, shlex, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
GZ = len(sys.argv) > 4 and sys.argv[4] == "gz"
L = 5_000_000
cli = os.environ.get("MIEKKI_CLI") or os.path.join(ROOT, "miekki_amd", "miekki")
# MIEKKI_PREFIX: words put before the binary ("rocprofv3 --kernel-trace -d DIR{rep} -o t --": {rep} = the run's number)
prefix = os.environ.get("MIEKKI_PREFIX", "")
def write_genome(job):
    g, fn = job
    data = synth.fasta(f"genome{g}", synth.genome_bases(g, 0, L))
    with open(fn, "wb") as f:
        f.write(gzip.compress(data, 6) if GZ else data)


with tempfile.TemporaryDirectory(prefix="mk_ing_", dir="/tmp") as d:
    t0 = time.time()
    ext = ".fa.gz" if GZ else ".fa"
    jobs = [(g, os.path.join(d, f"g{g}{ext}")) for g in range(min(D, G))]
    if len(jobs) > 128:                                 # (many distinct genomes: a collection whose Bloom filter fills up, as a real one's does)
        import multiprocessing
        with multiprocessing.Pool(min(16, os.cpu_count() or 1)) as pool:
            pool.map(write_genome, jobs, chunksize=8)
    else:
        for j in jobs:
            write_genome(j)
    with open(os.path.join(d, "genomes.lst"), "w") as lst:
        for g in range(G):
            fn = os.path.join(d, f"g{g}{ext}")
            if g >= D:
                os.symlink(os.path.join(d, f"g{g % D}{ext}"), fn)
            lst.write(fn + "\n")
    with open(os.path.join(d, "q.fa"), "wb") as f:
        f.write(b">q0\n" + synth.genome_bases(3, 1000, 1000) + b"\n")
    print(f"generated {D} distinct {'gzipped ' if GZ else ''}genomes ({G} listed) in {time.time() - t0:.1f}s", flush=True)
    # MIEKKI_VARIANTS="name:K=V,K=V;other:K=V": the same files under several environments, one after the other (A/B on one box)
    variants = [("", {})]
    if os.environ.get("MIEKKI_VARIANTS"):
        variants = [(v.split(":", 1)[0], dict(kv.split("=", 1) for kv in v.split(":", 1)[1].split(",") if kv)) for v in os.environ["MIEKKI_VARIANTS"].split(";")]
    for vname, venv, rep in [(a, b, r) for a, b in variants for r in range(2)]:
        if vname and rep == 0:
            print(f"# variant {vname}: {venv}", flush=True)
        time.sleep(float(os.environ.get("MIEKKI_PAUSE", "6")))      # (between runs: the driver is still taking back the tens of gigabytes of the process before -- device allocations of the next one then take seconds, profiles/r6_ingest_gz.txt)
        t0 = time.time()
        out = subprocess.run(shlex.split(prefix.replace("{rep}", str(rep))) + [cli, "-l", "genomes.lst", "-a", "q.fa", "-o", "out.txt", "-h", "20", "-t", str(T)], cwd=d, env=dict(os.environ, MIEKKI_VERBOSE="1", **venv),
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
        el = re.findall(r"elapsed time: ([0-9.e+-]+)s", out)
        wall = time.time() - t0
        idx = float(el[0]) if el else float("nan")
        print(f"run {rep}: wall {wall:.2f}s; index phase {idx:.2f}s = {G / idx:.0f} genomes/s, {G * L / idx / 1e9:.2f} GB/s of sequence", flush=True)
        for line in out.splitlines():
            for tag in ("[ingest]", "[gz]"):          # (the [gz] lines come on stderr, in the middle of the progress marks)
                if tag in line:
                    print("   ", line[line.index(tag):])
