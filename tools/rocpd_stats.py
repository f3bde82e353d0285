#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 run whose output
is a rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db on this image).  Prints CSV like rocprofv3's own kernel_stats.csv.
    python tools/rocpd_stats.py gpurun_out/prof/NAME_results.db > profiles/rN_xxx_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else "name")
q = f"""select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.{name_col} order by 3 desc"""
rows = list(cur.execute(q))
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, calls, tot, avg, mn, mx in rows:
    print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.3f},{mn},{mx}')
