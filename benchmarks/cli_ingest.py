#!/usr/bin/env python3
"""End-to-end time of the drop-in CLI on a tab-separated file (host parse + upload + GPU estimation),
beside the reference's row loop on a sample of the same text (the fallback path of the same CLI,
which is the reference's `csv.reader` + `int()` loop feeding `appendRow`).

  python benchmarks/cli_ingest.py [--rows 1000000] [--k 100] [--sample-rate 0.5] [--max-count 150]

--sample-rate / --max-count time the same file again with -s / -M (the fast parser with the reference's
draws from Python's `random` stream, findDirichletPrior._select_and_cap).
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
    ap.add_argument("--sample-rate", type=float, default=0.5)
    ap.add_argument("--max-count", type=int, default=150)
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

    import random
    timed = {}
    for name, extra in (("s", ["-s", str(args.sample_rate)]), ("M", ["-M", str(args.max_count)]),
                        ("s_M", ["-s", str(args.sample_rate), "-M", str(args.max_count)])):
        random.seed(1)
        t0 = time.perf_counter()
        cli.main(["-K", str(args.k), "-L", "ERROR"] + extra, stdin=_Stdin(raw), stdout=io.StringIO())
        timed[name] = time.perf_counter() - t0

    # the reference's row loop on a sample: a quoted first field is outside the fast parser's grammar
    sample_rows = min(args.rows, 20_000)
    cut = 0
    for _ in range(sample_rows):
        cut = raw.index(b"\n", cut) + 1
    t0 = time.perf_counter()
    cli.main(["-K", str(args.k), "-L", "ERROR"], stdin=_Stdin(b'"0"' + raw[raw.index(b"\t"):cut]),
             stdout=io.StringIO())
    t_loop = time.perf_counter() - t0

    print(json.dumps({
        "rows": args.rows, "K": args.k, "text_MB": len(raw) / 1e6, "generate_s": t_gen,
        "cli_seconds": t_fast, "cli_rows_per_s": args.rows / t_fast, "cli_text_MBps": len(raw) / 1e6 / t_fast,
        "cli_seconds_with_s_%g" % args.sample_rate: timed["s"],
        "cli_seconds_with_M_%d" % args.max_count: timed["M"],
        "cli_seconds_with_s_and_M": timed["s_M"],
        "row_loop_sample_rows": sample_rows, "row_loop_seconds": t_loop,
        "row_loop_rows_per_s": sample_rows / t_loop, "host_cpus_used": min(os.cpu_count() or 1, 16),
        "alpha_head": priors[:3]}))


if __name__ == "__main__":
    main()
