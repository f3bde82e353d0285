#!/usr/bin/env python3
"""End-to-end timing of the `miekki` host binary on synthetic FASTA files
(host parsing + PCIe included), for the host-inclusive note in DESIGN.md.
    python tools/cli_e2e.py [n_genomes] [n_queries] [threads]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
L = 5_000_000
cli = os.path.join(ROOT, "miekki_amd", "miekki")
with tempfile.TemporaryDirectory(prefix="mk_e2e_", dir="/tmp") as d:
    t0 = time.time()
    with open(os.path.join(d, "genomes.lst"), "w") as lst:
        for g in range(G):
            fn = os.path.join(d, f"g{g}.fa")
            with open(fn, "wb") as f:
                f.write(synth.fasta(f"genome{g}", synth.genome_bases(g, 0, L)))
            lst.write(fn + "\n")
    with open(os.path.join(d, "queries.fa"), "wb") as f:
        for q in range(Q):
            g, off = synth.query_origin(q, G, L, 1000)
            f.write(f">q{q}_g{g}\n".encode() + synth.genome_bases(g, off, 1000) + b"\n")
    print(f"generated {G} genomes + {Q} queries in {time.time() - t0:.1f}s", flush=True)
    for args, tag in ((["-l", "genomes.lst", "-a", "queries.fa", "-o", "out.txt", "-h", "20", "-t", str(T)], "build+query"),
                      (["-l", "genomes.lst", "-d", "idx.gz", "-o", "o2.txt", "-h", "20", "-t", str(T)], "build+dump"),
                      (["-i", "idx.gz", "-a", "queries.fa", "-o", "out_i.txt", "-t", str(T)], "load+query"),
                      (["-i", "idx.gz", "-A", "genomes.lst", "-o", "outA.txt", "-t", str(T)], "load+whole-genome queries"),
                      (["-l", "genomes.lst", "-a", "queries.fa", "-e", "-o", "exact.txt", "-h", "20", "-t", str(T)], "build+exact mode")):
        t0 = time.time()
        out = subprocess.run([cli, *args], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
        el = re.findall(r"elapsed time: ([0-9.e+-]+)s", out)
        print(f"{tag}: wall {time.time() - t0:.2f}s; phases (index, query) = {el}", flush=True)
    a = open(os.path.join(d, "out.txt"), "rb").read(); b = open(os.path.join(d, "out_i.txt"), "rb").read()
    ok = sum(1 for q, line in enumerate(a.split(b"\n")[:Q]) if line.split(b":")[1].split(b"\t")[0] == str(q % G).encode())
    print(f"-l and -i outputs identical: {a == b}; top hit = source genome for {ok}/{Q} queries; idx.gz = {os.path.getsize(os.path.join(d, 'idx.gz')) / 1e6:.0f} MB")
    nA = len(open(os.path.join(d, "outA.txt"), "rb").read().splitlines())
    print(f"-A lines: {nA}")
    ex = open(os.path.join(d, "exact.txt"), "rb").read().splitlines()
    good = sum(1 for l in ex if float(l.split(b"\t")[2]) >= 900)
    print(f"exact-mode lines: {len(ex)}; with >= 900 shared k-mers (the source genome): {good}")
