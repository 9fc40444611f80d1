#!/usr/bin/env python3
"""Prints the kernel timeline of one bench step from a rocprofv3 --kernel-trace database
(rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...  ->  DIR/NAME_results.db)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select d.start,d.end,s.kernel_name,d.grid_size_x,d.workgroup_size_x from {kd} d join {ks} s "
                        f"on d.kernel_id=s.id order by d.start"))
key = sys.argv[2] if len(sys.argv) > 2 else "stats_dense_kernel"
idx = [i for i, r in enumerate(rows) if key in r[2]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
i0, i1 = idx[-back], idx[-back + 1]
t0 = rows[i0][0]
prev_end = None
for r in rows[i0 - 2:i1 + 1]:
    gap = (r[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f dur %8.1f gap %6.1f  grid %7d x %4d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap,
                                                             r[3], r[4], r[2][:70]))
    prev_end = r[1]
