#!/usr/bin/env python3
"""Merge one configuration's scan traffic into profiles/pmc_traffic.json from two rocprofv3 counter passes of bench.py
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with --kernel-trace --output-format csv):
    python tools/pmc_traffic.py --genomes G --queries Q --h H --fp-bits B --source "text" FETCH.csv WRITE.csv [--launches-per-step N]
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a 16 B/lane streaming read;
Infinity-Cache hits are counted too, MI355X_MICROARCH.md section HBM), averaged over all launches of the scan kernel."""
import argparse
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("fetch"); ap.add_argument("write")
ap.add_argument("--genomes", type=int, required=True); ap.add_argument("--queries", type=int, required=True)
ap.add_argument("--h", type=int, required=True); ap.add_argument("--fp-bits", type=int, required=True)
ap.add_argument("--source", default=""); ap.add_argument("--kernel", default="scan_slab_kernel")
ap.add_argument("--steps-counted", type=int, default=2, help="steps the counter passes ran (--warmup 1 --steps 1: two)")
a = ap.parse_args()


def vals(path, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and a.kernel in r["Kernel_Name"]]


f, w = vals(a.fetch, "FETCH_SIZE"), vals(a.write, "WRITE_SIZE")
assert f and w, "no launches of %s in the counter files" % a.kernel
entry = {"genomes_per_gpu": a.genomes, "queries": a.queries, "h": a.h, "fp_bits": a.fp_bits, "kernel": a.kernel,
         "traffic_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
         "fetch_size_kb_mean": sum(f) / len(f), "write_size_kb_mean": sum(w) / len(w), "launches_counted": len(f), "launches_per_step": len(f) / a.steps_counted,
         "traffic_bytes_per_step": (2 * sum(f) + sum(w)) * 1024 / a.steps_counted, "source": a.source}
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pm = json.load(open(path))
pm["entries"] = [e for e in pm["entries"] if (e["genomes_per_gpu"], e["queries"], e["h"], e["fp_bits"]) != (a.genomes, a.queries, a.h, a.fp_bits)] + [entry]
json.dump(pm, open(path, "w"), indent=1)
print(json.dumps(entry))
