#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 `--pmc ... --output-format csv` run.
    python tools/pmc_summary.py DIR/NAME_counter_collection.csv [kernel-name-substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if want not in k:
        continue
    print(k[:90])
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):5d} mean={sum(v) / len(v):16.1f}")
