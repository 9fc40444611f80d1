#!/usr/bin/env python3
"""BASELINE configs[4]: heavy-tailed counts, K = 100, N = 1 000 000 int32 dense rows,
row totals min(floor(5 * (Pareto(1.2) + 1)), 100 000).  U/V reach 1e5 entries per column, so one
solver evaluation streams tens of MB: this is the one configuration where the evaluation kernel
has a bandwidth figure (B_eval = (sum lenU + lenV) * 8 bytes).

  python benchmarks/heavy_tail.py [--rows 1000000] [--steps 5]
Prints one JSON line: statistics pass ms / GB/s, evaluation us / GB/s, solve ms, iterations.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from bayespy_amd import device as dev

    N, K = args.rows, args.k
    rng = np.random.default_rng(20260005)
    alpha_true = 0.05 + rng.gamma(2.0, 0.3, size=K)
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.rand(N, device="cuda", dtype=torch.float64, generator=g)
    totals = torch.clamp(torch.floor(5.0 * (1.0 - u) ** (-1.0 / 1.2)), max=100000).to(torch.int32)
    X = dev.synth_dirmult(N, K, alpha_true, totals, seed=20260005, dtype=torch.int32)
    max_total = int(totals.max())
    max_count = int(X.max())
    ev = lambda: torch.cuda.Event(enable_timing=True)

    stats_ms, eval_us, solve_ms = [], [], []
    for it in range(args.steps + 1):
        st = dev.DeviceStats(K, max_count, max_total)
        e0, e1 = ev(), ev()
        e0.record()
        st.add_dense(X, max_count=max_count, max_total=max_total)
        e1.record()
        uv = st.finalize()
        a0 = st.cli_init_alphas()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = uv.solve(a0, 50, 0.0, trace_alphas=False)
        t1 = time.perf_counter()
        # one evaluation (segment table + per-segment sums + the single-workgroup gather)
        e2, e3 = ev(), ev()
        e2.record()
        for _ in range(10):
            uv.evaluate(out["alpha_device"].cpu().numpy(), 0.0)
        e3.record()
        torch.cuda.synchronize()
        if it:
            stats_ms.append(e0.elapsed_time(e1))
            solve_ms.append((t1 - t0) * 1e3)
            eval_us.append(e2.elapsed_time(e3) * 100.0)
    info = st.check()
    lenU = uv.lenU.cpu().numpy()
    lenV = int(uv.lenV.cpu()[0])
    b_eval = (int(lenU.sum()) + lenV) * 8
    b_stats = N * K * 4
    line = {"workload": "heavy-tailed counts (BASELINE configs[4])", "K": K, "rows": N,
            "max_total": info["max_total"], "max_count": info["max_count"],
            "sum_lenU": int(lenU.sum()), "lenV": lenV,
            "layout": "ragged" if uv.ragged is not None else "dense-padded",
            "U_terms_allocated": int(uv.Uf.numel()), "dense_cells_K_x_max_count": K * info["max_count"],
            "stats_ms": float(np.mean(stats_ms)), "stats_GBps": b_stats / (np.mean(stats_ms) * 1e-3) / 1e9,
            "solve_ms": float(np.mean(solve_ms)), "iterations": out["n_iters"], "loss_evals": out["n_evals"],
            "launches": out["launches"], "exit": out["exit"],
            "evaluate_call_us_incl_host": float(np.mean(eval_us)), "B_eval_bytes": b_eval,
            "solve_GBps_over_evals": b_eval * out["n_evals"] / (np.mean(solve_ms) * 1e-3) / 1e9}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
