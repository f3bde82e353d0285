#!/usr/bin/env python3
"""Start / end of the kernel dispatches of a rocprofv3 run (rocpd SQLite database), in microseconds from the first one
shown -- to see which kernels of two streams actually ran side by side.
    python tools/rocpd_timeline.py DIR/NAME_results.db [first dispatch] [how many]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cur = db.cursor()
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else "name")
rows = list(cur.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
                            join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start limit ? offset ?""", (count, first)))
rows = [(re.search(r"(\w+_kernel|__amd_rocclr_\w+)", n).group(1) if re.search(r"(\w+_kernel|__amd_rocclr_\w+)", n) else n[:40], a, b)
        for n, a, b in rows]
# memory copies (rocprofv3 --memory-copy-trace), when the run recorded them: those inside the window of the kernels shown
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view') and name like '%memory_copy%'")]
for t in tables:
    cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
    if "start" in cols and "end" in cols and rows:
        size = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
        lo, hi = rows[0][1], rows[-1][2]
        q = f"select {size or 0}, start, end from {t} where end >= ? and start <= ? order by start"
        rows += [(f"copy {sz} B", a, b) for sz, a, b in cur.execute(q, (lo, hi))]
        break
rows.sort(key=lambda r: r[1])
if rows:
    t0 = rows[0][1]
    for name, a, b in rows:
        print(f"{(a - t0) / 1e3:10.1f} {(b - t0) / 1e3:10.1f} {(b - a) / 1e3:9.1f}  {name}")
