#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc databases of `bench.py` (one pass each for FETCH_SIZE
and WRITE_SIZE, as MI355X_MICROARCH.md prescribes): the HBM bytes of one full-size launch of the
statistics kernel, tied to the source it was measured on.

  traffic_json.py FETCH_DB WRITE_DB COMMIT [N] [K] [dtype] > profiles/traffic.json

bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (the guide's gfx950 correction: FETCH_SIZE reports half
the bytes of a wide coalesced streaming read; WRITE_SIZE is exact for 16-byte stores).  Launches that
fetched less than half of the largest one (bench.py's 4096-row initialisation) are left out.
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(dbpath, counter, name="stats_dense_kernel"):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    kd, ks, pe, pi = (t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"),
                      t("rocpd_info_pmc"))
    q = (f"select d.id, sum(e.value) from {pe} e join {pi} i on e.pmc_id=i.id join {kd} d on "
         f"e.event_id=d.event_id join {ks} s on d.kernel_id=s.id where i.name=? and s.kernel_name like ? "
         f"group by d.id")
    return [v for _, v in cur.execute(q, (counter, "%" + name + "%"))]


def blob(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def main():
    fdb, wdb, commit = sys.argv[1:4]
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
    K = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
    dtype = sys.argv[6] if len(sys.argv) > 6 else "u16"
    f = per_launch(fdb, "FETCH_SIZE")
    w = per_launch(wdb, "WRITE_SIZE")
    fbig = [x for x in f if x > 0.5 * max(f)]
    wbig = sorted(w)[-len(fbig):] if len(w) >= len(fbig) else w
    fk, wk = sum(fbig) / len(fbig), sum(wbig) / len(wbig)
    out = {
        f"stats_dense_{dtype}_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py "
               "--steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0` (benchmarks/pmc_bench_traffic.sh); "
               "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE "
               "reports exactly half of the bytes of a wide coalesced streaming read; WRITE_SIZE is exact "
               "for 16-byte stores); mean over the %d full-size launches of stats_dense_kernel" % len(fbig),
        "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "full_size_launches": len(fbig),
        "algorithmic_bytes_per_launch": float(N * K * (2 if dtype == "u16" else 4)),
        "N": N, "K": K,
        "dme_stats_hip_blob": blob(os.path.join(ROOT, "bayespy_amd", "csrc", "dme_stats.hip")),
        "dme_common_h_blob": blob(os.path.join(ROOT, "bayespy_amd", "csrc", "dme_common.h")),
        "commit": commit,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
