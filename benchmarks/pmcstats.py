#!/usr/bin/env python3
"""Per-kernel PMC counter sums of a rocprofv3 --pmc database: pmcstats.py DB [name-filter]."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols_pe = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
cols_pi = [r[1] for r in cur.execute(f"pragma table_info({pi})")]
q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pe} e join {pi} i on e.pmc_id=i.id "
     f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name")
acc = defaultdict(dict)
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("query failed:", ex, cols_pe, cols_pi)
    raise
for name, cn, val, n in rows:
    if flt in name:
        acc[name][cn] = (val, n)
for name, d in acc.items():
    print(name[:80])
    for cn, (val, n) in sorted(d.items()):
        print("    %-28s %16.0f per launch (%d launches)" % (cn, val / n, n))
