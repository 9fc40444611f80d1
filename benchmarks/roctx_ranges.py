#!/usr/bin/env python3
"""The roctx ranges of a `rocprofv3 --marker-trace` database: roctx_ranges.py DB  (count / mean us per name).
rocprofv3 7.2 stores a range as a row of rocpd_region (start, end) whose event's extdata holds
{"message": "<the range's name>"}."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
reg = [t for t in tabs if t.startswith("rocpd_region")]
ev = [t for t in tabs if t.startswith("rocpd_event")]
if not reg or not ev:
    print("no region / event table; tables:", ", ".join(tabs))
    sys.exit(0)
acc = {}
for start, end, ext in cur.execute(f"select r.start, r.end, e.extdata from {reg[0]} r join {ev[0]} e on r.event_id = e.id"):
    try:
        name = json.loads(ext).get("message", "")
    except (TypeError, ValueError):
        continue
    if name.startswith("dme:"):
        n, t = acc.get(name, (0, 0.0))
        acc[name] = (n + 1, t + (end - start) / 1e3)
for name in sorted(acc):
    n, t = acc[name]
    print("%-16s n=%4d  host side of the phase, mean %9.1f us" % (name, n, t / n))
if not acc:
    print("no dme: ranges in the trace (DME_ROCTX=1 not set?)")
