#!/usr/bin/env python3
"""Per-kernel time summary of a rocprofv3 --kernel-trace database: kstats.py DB [name-filter]."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = (f"select s.kernel_name,count(*),avg(d.end-d.start)/1e3,min(d.end-d.start)/1e3,max(d.end-d.start)/1e3 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
for r in cur.execute(q):
    if flt in r[0]:
        print("%-88s n=%5d avg %9.1f us  min %9.1f  max %9.1f" % (r[0][:88], r[1], r[2], r[3], r[4]))
