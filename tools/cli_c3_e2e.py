#!/usr/bin/env python3
"""The drop-in as its user sees it at the metric's size (BASELINE config 3: 100,000 genomes at -h 20; main.cpp:209-211,
232-234 print the two elapsed times this reads): the `miekki` binary over a 100,000-genome index FILE --
    -i idx -a queries.fa      100,000 x 1 kb queries (load, then the query phase: parse, sketch, scan, heap, out.txt)
    -i idx -A wholes.lst      64 whole 5 Mb genomes as queries
-- and, because exact mode needs the genome FILES (-l; the index file does not keep their names, Miekki.h:59), BASELINE
config 5's shape at -h 20 so that there are hits:
    -l 1000 genomes -a 20,000 queries -e
The index file is made by tools/index_io_bench (synthetic genomes built on the device, dumped with the binary's own
writer) into /dev/shm.   python tools/cli_c3_e2e.py [genomes] [queries] [threads]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
L = 5_000_000
cli = os.path.join(ROOT, "miekki_amd", "miekki")
iob = os.path.join(ROOT, "tools", "index_io_bench")


def write_genome(job):
    g, fn = job
    open(fn, "wb").write(synth.fasta(f"genome{g}", synth.genome_bases(g, 0, L)))
    return fn


def run(args, cwd, tag):
    t0 = time.time()
    out = subprocess.run([cli, *args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, MIEKKI_DEVICES="0")).stdout.decode()
    el = re.findall(r"elapsed time: ([0-9.e+-]+)s", out)
    print(f"{tag}: wall {time.time() - t0:.2f}s; the binary's own elapsed times: index/load {el[0] if el else '?'} s, queries {el[1] if len(el) > 1 else '?'} s", flush=True)
    return out


with tempfile.TemporaryDirectory(prefix="mk_c3_", dir="/dev/shm") as d:
    idx = os.path.join(d, "idx.gz")
    t0 = time.time()
    r = subprocess.run([iob, str(G), "20", "8", str(T), idx, "keep"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print(r.stdout.decode().strip(), flush=True)
    assert r.returncode == 0
    print(f"index file: {os.path.getsize(idx) / 1e9:.1f} GB in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    with open(os.path.join(d, "queries.fa"), "wb") as f:
        for q in range(Q):
            g, off = synth.query_origin(q, G, L, 1000)
            f.write(f">q{q}_g{g}\n".encode() + synth.genome_bases(g, off, 1000) + b"\n")
    from multiprocessing import Pool
    pool = Pool(min(T, 16))
    with open(os.path.join(d, "wholes.lst"), "w") as lst:
        for fn in pool.map(write_genome, [((j * 1543) % G, os.path.join(d, f"w{j}.fa")) for j in range(64)]):
            lst.write(fn + "\n")
    print(f"{Q} queries and 64 whole genomes written in {time.time() - t0:.0f} s", flush=True)
    run(["-i", idx, "-a", "queries.fa", "-o", "out.txt", "-t", str(T)], d, f"-i (100,000 genomes) -a ({Q} x 1 kb)")
    lines = open(os.path.join(d, "out.txt"), "rb").read().split(b"\n")
    ok = sum(1 for q, line in enumerate(lines[:Q]) if b":" in line and line.split(b":")[1].split(b"\t")[0] == str(q % G).encode())
    print(f"    top hit = source genome for {ok}/{Q} queries")
    run(["-i", idx, "-a", "queries.fa", "-o", "out2.txt", "-t", str(T)], d, "the same again (file in the page cache, warm start)")
    run(["-i", idx, "-A", "wholes.lst", "-o", "outA.txt", "-t", str(T)], d, "-i -A (64 whole 5 Mb genomes)")
    la = open(os.path.join(d, "outA.txt"), "rb").read().splitlines()
    okA = sum(1 for j, line in enumerate(la) if b":" in line and line.split(b":")[1].split(b"\t")[0] == str((j * 1543) % G).encode())
    print(f"    top hit = the genome itself for {okA}/64 files")
    os.remove(idx)
    # exact mode (config 5's shape at -h 20)
    Ge, Qe = 1000, 20000
    t0 = time.time()
    with open(os.path.join(d, "genomes.lst"), "w") as lst:
        for fn in pool.map(write_genome, [(g, os.path.join(d, f"g{g}.fa")) for g in range(Ge)]):
            lst.write(fn + "\n")
    with open(os.path.join(d, "qe.fa"), "wb") as f:
        for q in range(Qe):
            g, off = synth.query_origin(q, Ge, L, 1000)
            f.write(f">q{q}_g{g}\n".encode() + synth.genome_bases(g, off, 1000) + b"\n")
    print(f"{Ge} genome files and {Qe} queries written in {time.time() - t0:.0f} s", flush=True)
    run(["-l", "genomes.lst", "-a", "qe.fa", "-o", "approx.txt", "-h", "20", "-t", str(T)], d, f"-l ({Ge} x 5 Mb files) -a ({Qe} x 1 kb)")
    run(["-l", "genomes.lst", "-a", "qe.fa", "-e", "-o", "exact.txt", "-h", "20", "-t", str(T)], d, "the same with -e (exact mode)")
    ex = open(os.path.join(d, "exact.txt"), "rb").read().splitlines()
    print(f"    exact.txt: {len(ex)} lines")
