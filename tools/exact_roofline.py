#!/usr/bin/env python3
"""Exact mode's bench line with a roofline: joins tools/bench_exact.py's JSON line with the kernel statistics of the same
command under `rocprofv3 --kernel-trace --stats` (tools/rocpd_stats.py) and, when given, `--pmc FETCH_SIZE` / `--pmc
WRITE_SIZE` passes (counter_collection.csv).
    python tools/exact_roofline.py BENCH.json KERNEL_STATS.csv [FETCH.csv WRITE.csv] > profiles/rN_bench_exact.json

What bounds K7 (exact_kernel<build>): every k-mer of the genome is one 64-bit compare-and-swap at a random slot of an
open-addressing set (2^24 slots x 8 B = 128 MiB for a 5 Mb genome: it fits the 256 MiB Infinity Cache) -- one 64-byte
line read and written back per k-mer, whatever the 8 bytes that matter.  Algorithmic bytes per genome = the characters
(1 B per k-mer) + 2 x 64 B per k-mer of set traffic; `achieved` = those over the kernel's average duration, `peak` the
8 TB/s of MI355X_MICROARCH.md; `traffic` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch when counters are given."""
import csv
import json
import sys

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
stats = {}
for r in csv.DictReader(open(sys.argv[2])):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
build = next((v for k, v in stats.items() if "exact_kernel" in k and "ILb0E" in k), None)
query = next((v for k, v in stats.items() if "exact_kernel" in k and "ILb1E" in k), None)
Lg, K = 5_000_000, 31
kmers = Lg - K + 1
algo = kmers * (1 + 128)
traffic = None
if len(sys.argv) > 4:
    def mean(path, counter):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "exact_kernel" in r["Kernel_Name"] and ("ILb0E" in r["Kernel_Name"] or "<false>" in r["Kernel_Name"])]
        return sum(v) / len(v) if v else None
    f, w = mean(sys.argv[3], "FETCH_SIZE"), mean(sys.argv[4], "WRITE_SIZE")
    if f is not None and w is not None:
        traffic = (2 * f + w) * 1024
if build:
    s = build[1] / 1e9
    line["roofline"] = {"kernel": "exact_kernel<build>", "bound": "hbm", "bound_detail": "one random 64-byte line read and written per k-mer (64-bit atomicCAS into a 128 MiB open-addressing set: Infinity-Cache resident)",
                        "achieved": algo / s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": algo / s / 1e9 / 8000.0, "traffic": traffic,
                        "traffic_note": "the counters see the genome's characters as fetches (2 x 2.7 MB) and every atomic as one 64-byte WRITE (379 MB for 5.0 M k-mers, duplicates included): device-scope atomics are executed memory-side, the line never comes to the L2 -- so the measured traffic is the model's write half",
                        "algorithmic_bytes_per_launch": algo, "avg_launch_ms": build[1] / 1e6, "launches": build[0], "kmers_per_s": kmers / s,
                        "atomics_per_s": kmers / s}
if query:
    line["k7_query_kernel_avg_ms"] = query[1] / 1e6
print(json.dumps(line))
