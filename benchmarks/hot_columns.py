#!/usr/bin/env python3
"""The dense statistics pass on data whose counts do NOT fit the packed LDS bins (VERDICT r1, weak
item 3): the shape of BASELINE configs[2] (K = 1000, row totals 1-500, uint16) with most of the
Dirichlet mass on a few hot columns, beside the bench's own friendly distribution.

  alpha_friendly = 0.01 + Gamma(2, 0.05)                       (bench.py, numpy seed 0; sum ~ 110)
  alpha_hot      = the same vector with `--hot` columns raised so that they hold `--mass` of the sum,
                   then scaled to sum `--sum` (a small sum concentrates every row in a few of them)

  python benchmarks/hot_columns.py [--rows 10000000] [--hot 10] [--mass 0.9] [--steps 10]
Prints one JSON line: ms and GB/s of the statistics launch (HIP events) for both distributions, the
share of cells above the packed bins' range, and the integer checks (column sums, row count).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--max-total", type=int, default=500)
    ap.add_argument("--hot", type=int, default=10)
    ap.add_argument("--mass", type=float, default=0.9)
    ap.add_argument("--sum", type=float, default=3.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", choices=["u16", "i32"], default="u16")
    ap.add_argument("--lib", default=None, help="another build of libdme_hip.so to time (A/B comparisons)")
    args = ap.parse_args()
    import torch
    from bayespy_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from bayespy_amd import device as dev

    K, N, M = args.k, args.rows, args.max_total
    rng = np.random.default_rng(0)
    friendly = 0.01 + rng.gamma(2.0, 0.05, size=K)
    hot = friendly.copy()
    cols = np.random.default_rng(1).choice(K, size=args.hot, replace=False)
    rest = hot.sum() - hot[cols].sum()
    hot[cols] = (args.mass / (1.0 - args.mass)) * rest / args.hot
    hot *= args.sum / hot.sum()
    tdtype = torch.uint16 if args.dtype == "u16" else torch.int32
    esize = 2 if args.dtype == "u16" else 4
    gen = torch.Generator(device="cuda").manual_seed(20260003)
    totals = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda", generator=gen)
    out = {"workload": "K=%d, %d rows, row totals 1-%d, %s dense" % (K, N, M, args.dtype),
           "hot_columns": args.hot, "hot_mass": args.mass, "hot_alpha_sum": args.sum}
    X = torch.empty((N, K), dtype=tdtype, device="cuda")
    for name, alpha in (("friendly", friendly), ("hot", hot)):
        dev.synth_dirmult(N, K, alpha, totals, seed=20260003, dtype=tdtype, out=X)
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
        assert info["rows"] == N
        uv = st.finalize(ragged=False)
        U = uv.Ui.view(K, uv.ldU)
        colsum = torch.zeros(K, dtype=torch.int64, device="cuda")
        big = 0
        for lo in range(0, N, 500_000):
            blk = X[lo:lo + 500_000]
            blk = blk.view(torch.int16).to(torch.int64).bitwise_and(0xFFFF) if esize == 2 else blk.to(torch.int64)
            colsum += blk.sum(dim=0)
            big += int((blk > 64).sum())
        assert torch.equal(U.sum(dim=1), colsum), "column sums"
        assert int(uv.Vi.sum()) == int(totals.to(torch.int64).sum())
        t = float(np.median(ms))
        out[name] = {"stats_ms": t, "GBps": N * K * esize / t / 1e6, "frac_of_8TBps": N * K * esize / t / 1e6 / 8000.0,
                     "max_count": info["max_count"], "cells_above_64_per_row": big / N}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
