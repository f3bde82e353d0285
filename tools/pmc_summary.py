#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 `--pmc ... --output-format csv` run.
    python tools/pmc_summary.py DIR/NAME_counter_collection.csv [kernel-name-substring] [skip]
skip: leave out the first `skip` launches of each kernel (e.g. the batches of a build whose Bloom filter is still young)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    rows = list(csv.DictReader(f))
if rows and "Dispatch_Id" in rows[0]:
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2] if len(sys.argv) > 2 else ""
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for k, cs in acc.items():
    if want not in k:
        continue
    print(k[:90] + (f"   (launches {skip}.. of each counter)" if skip else ""))
    for c, v in sorted(cs.items()):
        v = v[skip:] or v
        print(f"   {c:28s} n={len(v):5d} mean={sum(v) / len(v):16.1f}")
