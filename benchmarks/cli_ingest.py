#!/usr/bin/env python3
"""End-to-end time of the drop-in CLI on a tab-separated file (host parse + upload + GPU estimation),
beside the reference's row loop on a sample of the same text (the fallback path of the same CLI,
which is the reference's `csv.reader` + `int()` loop feeding `appendRow`).

  python benchmarks/cli_ingest.py [--rows 1000000] [--k 100]
"""
import argparse
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _Stdin(raw):
    """A text stream with a .buffer, like sys.stdin."""
    return io.TextIOWrapper(io.BytesIO(raw))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=100)
    args = ap.parse_args()
    from bayespy_amd import findDirichletPrior as cli

    rng = np.random.default_rng(7)
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=args.k)
    t0 = time.time()
    lines = []
    for lo in range(0, args.rows, 50_000):
        n = min(50_000, args.rows - lo)
        p = rng.dirichlet(alpha, size=n)
        tot = rng.integers(1, 200, size=n)
        X = np.stack([rng.multinomial(int(t), q) for t, q in zip(tot, p)])
        lines.extend("\t".join(map(str, r)) for r in X.tolist())
    raw = ("\n".join(lines) + "\n").encode()
    del lines
    t_gen = time.time() - t0

    cli.main(["-K", str(args.k), "-L", "ERROR"], stdin=_Stdin(raw[:raw.index(b"\n", len(raw) // 50) + 1]),
             stdout=io.StringIO())  # warm-up (library load, first launches)
    t0 = time.perf_counter()
    priors = cli.main(["-K", str(args.k), "-L", "ERROR"], stdin=_Stdin(raw), stdout=io.StringIO())
    t_fast = time.perf_counter() - t0

    # the reference's row loop on a sample: -M with a cap nobody reaches switches the fast parser off
    sample_rows = min(args.rows, 20_000)
    cut = 0
    for _ in range(sample_rows):
        cut = raw.index(b"\n", cut) + 1
    t0 = time.perf_counter()
    cli.main(["-K", str(args.k), "-L", "ERROR", "-M", str(2**40)], stdin=_Stdin(raw[:cut]), stdout=io.StringIO())
    t_loop = time.perf_counter() - t0

    print(json.dumps({
        "rows": args.rows, "K": args.k, "text_MB": len(raw) / 1e6, "generate_s": t_gen,
        "cli_seconds": t_fast, "cli_rows_per_s": args.rows / t_fast, "cli_text_MBps": len(raw) / 1e6 / t_fast,
        "row_loop_sample_rows": sample_rows, "row_loop_seconds": t_loop,
        "row_loop_rows_per_s": sample_rows / t_loop, "host_cpus_used": min(os.cpu_count() or 1, 16),
        "alpha_head": priors[:3]}))


if __name__ == "__main__":
    main()
