#!/usr/bin/env python3
"""Weighted statistics pass at the size of a mixture-EM E-step (not a bench.py
line): N rows x K columns of int32 counts, C responsibilities per row, C fp64 histograms from one
read of X (dme_stats_dense_weighted_batch_i32), then the C weighted suffix scans.

  python benchmarks/weighted_em.py [--rows 1000000] [--k 100] [--c 5] [--max-total 100]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--c", type=int, default=5)
    ap.add_argument("--max-total", type=int, default=100)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from bayespy_amd import device as dev
    N, K, C, M = args.rows, args.k, args.c, args.max_total
    device = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    gen = torch.Generator(device=device).manual_seed(7)
    totals = torch.randint(1, M + 1, (N,), dtype=torch.int32, device=device, generator=gen)
    X = dev.synth_dirmult(N, K, alpha, totals, seed=7, dtype=torch.int32, device=device)
    R = torch.rand((N, C), dtype=torch.float64, device=device, generator=gen)
    R /= R.sum(dim=1, keepdim=True)
    torch.cuda.synchronize()
    times, scan_times = [], []
    for it in range(args.steps + 2):
        ws = dev.WeightedStats(K, C=C, device=device)
        ws.reserve(M, M)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ws.add_dense(X, R, M, M)
        e1.record()
        uvs = ws.to_uvs()
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
            scan_times.append(e1.elapsed_time(e2))
    ws.check()
    # the responsibilities of a row sum to 1: the C row-total histograms add up to the row count
    rows_seen = float(sum(float(u.Vf[0]) for u in uvs))
    ms = float(np.median(times))
    nbytes = N * K * 4 + N * C * 8
    print(json.dumps({
        "workload": "weighted rows, one E-step of a Dirichlet-mixture EM", "rows": N, "K": K, "C": C,
        "max_total": M, "algorithmic_bytes": nbytes, "stats_ms": ms, "stats_ms_min": float(min(times)),
        "achieved_GBps": nbytes / ms / 1e6, "rows_per_s": N / ms * 1e3,
        "scan_ms_for_C_components": float(np.median(scan_times)),
        "sum_of_V0_over_components": rows_seen}))


if __name__ == "__main__":
    main()
