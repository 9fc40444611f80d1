#!/usr/bin/env python3
"""The dense statistics pass across row widths and element types: ms and TB/s
of the statistics launch (HIP events around dme_stats_dense_*; median of `--steps` launches) for a
list of shapes, each on `--gb` GB of synthetic Dirichlet-multinomial rows (alpha = 0.01 +
Gamma(2, 0.05), numpy seed 0; row totals 1..max_total).

  python benchmarks/dense_shapes.py [--gb 8] [--steps 7] [--shapes 1000:u16,1000:i32,4096:u16,...]
A shape is K:dtype[:max_total].  Prints one JSON line per shape and a markdown table at the end.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULT = ("2:u16:60,4:u16:60,8:u16:100,16:u16:100,32:u16:100,64:u16:100,104:u16:100,200:u16:100,320:u16:200,"
           "400:u16:200,512:u16,600:u16,768:u16,1000:u16,1000:i32,1001:u16,1536:u16,2048:u16,2048:i32,2300:u16,"
           "2560:u16,3000:u16,4096:u16,4096:i32,8192:u16,16384:u16,32768:u16,8192:i32,16384:i32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--lib", default=None, help="another build of libdme_hip.so to time")
    ap.add_argument("--no-check", action="store_true", help="ablation builds: skip the row-count checks")
    args = ap.parse_args()
    import torch
    from bayespy_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from bayespy_amd import device as dev

    rows = []
    for spec in args.shapes.split(","):
        parts = spec.split(":")
        K, dt = int(parts[0]), parts[1]
        M = int(parts[2]) if len(parts) > 2 else 500
        esize = 2 if dt == "u16" else 4
        tdtype = torch.uint16 if dt == "u16" else torch.int32
        N = int(args.gb * 1e9 / (K * esize))
        rng = np.random.default_rng(0)
        alpha = 0.01 + rng.gamma(2.0, 0.05, size=K)
        gen = torch.Generator(device="cuda").manual_seed(20260003)
        totals = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda", generator=gen)
        X = dev.synth_dirmult(N, K, alpha, totals, seed=20260003, dtype=tdtype)
        ms = []
        for step in range(args.steps + 2):
            st = dev.DeviceStats(K, M, M)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.add_dense(X, max_count=M, max_total=M)
            e1.record()
            info = st.check()
            if step >= 2:
                ms.append(e0.elapsed_time(e1))
        if not args.no_check:
            assert info["rows"] == N
            assert int(st.histV.sum()) == N  # every row has a total >= 1
        t = float(np.median(ms))
        rec = {"K": K, "dtype": dt, "rows": N, "max_total": M, "bytes": N * K * esize, "stats_ms": t,
               "TBps": N * K * esize / t / 1e9, "frac_of_8TBps": N * K * esize / t / 1e9 / 8.0}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del X, totals, st
        torch.cuda.empty_cache()
    print("| K | dtype | rows | GB | ms | TB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %s | %d | %.1f | %.3f | %.2f | %.0f |" % (r["K"], r["dtype"], r["rows"], r["bytes"] / 1e9,
                                                                 r["stats_ms"], r["TBps"], 100 * r["frac_of_8TBps"]))


if __name__ == "__main__":
    main()
