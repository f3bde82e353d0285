#!/usr/bin/env python3
"""profiles/pmc_build.json: the DYNAMIC instruction counts of the index build's two big kernels, per configuration, from
`rocprofv3 --pmc SQ_* --kernel-trace --output-format csv` passes over tools/build_rate.py (under counter collection the
kernels run one at a time: their durations in the same run's kernel trace are "alone on the chip" times).

    python tools/pmc_build.py --h 20 --fp-bits 8 --source "text" COUNTERS.csv KERNEL_TRACE.csv [--skip 70]

Merges the entry for (k = 31, h, fp_bits) into profiles/pmc_build.json together with the SHA-256 of build.hip as it is NOW:
bench.py reports `sketch.roofline.frac` only from an entry whose configuration AND source hash match the run (ADVICE r3:
counters of another kernel generation or another -h must not be passed off as this run's)."""
import argparse
import csv
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("counters"); ap.add_argument("trace")
ap.add_argument("--h", type=int, required=True); ap.add_argument("--fp-bits", type=int, required=True)
ap.add_argument("--skip", type=int, default=70, help="leave out the first launches of each kernel (a young Bloom filter)")
ap.add_argument("--source", default="")
a = ap.parse_args()


def short(name):
    for k in ("build_scatter_kernel", "build_reduce_kernel"):
        if k in name:
            w = "2" if ("<2," in name or "ILi2E" in name) else "1"
            return f"{k}<{w}>"
    return None


acc = defaultdict(lambda: defaultdict(list))
rows = list(csv.DictReader(open(a.counters)))
if rows and "Dispatch_Id" in rows[0]:
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    k = short(r["Kernel_Name"])
    if k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
trows = list(csv.DictReader(open(a.trace)))
trows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in trows:
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
kernels = {}
for k, cs in acc.items():
    mean = lambda v: sum(v[a.skip:] or v) / len(v[a.skip:] or v)
    d = dur.get(k, [])
    kernels[k] = {"valu_wave_instructions_per_batch": mean(cs["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in cs else None,
                  "counters": {c: mean(v) for c, v in sorted(cs.items())},
                  "alone_ms": (sum(d[a.skip:] or d) / len(d[a.skip:] or d)) if d else None, "launches": len(d)}
entry = {"k": 31, "h": a.h, "fp_bits": a.fp_bits, "batch_genomes": 64, "genome_len": 5_000_000,
         "build_hip_sha256": hashlib.sha256(open(os.path.join(ROOT, "miekki_amd", "csrc", "build.hip"), "rb").read()).hexdigest(),
         "kernels": kernels, "source": a.source, "skipped_launches": a.skip}
path = os.path.join(ROOT, "profiles", "pmc_build.json")
try:
    doc = json.load(open(path))
    if "entries" not in doc:
        doc = {"entries": {}}
except Exception:
    doc = {"entries": {}}
doc["entries"][f"k31_h{a.h}_fp{a.fp_bits}"] = entry
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
