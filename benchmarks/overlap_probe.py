#!/usr/bin/env python3
"""What would it buy to run the solver of one estimation BESIDE the statistics pass of the next?

The solver's launches are small and latency-bound (251 workgroups, ~11 us each, 12 in a row) and the
statistics kernel cannot share a CU with them (it holds 156 of the 160 KB of LDS).  This probe gives
each its own CU-masked stream (dme_stream_create_masked: the solver `--solver-cus` compute units, the
statistics pass the rest; dme_set_cu_limit sizes the launches accordingly) and times K estimations
enqueued back to back, against the same K estimations on one unmasked stream (what bench.py does).

  python benchmarks/overlap_probe.py [--rows 1250000] [--solver-cus 16] [--steps 20]

--rows 1250000 is one GPU's share of BASELINE configs[2] at 8 GPUs; 10000000 is the bench itself.
Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_250_000)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--max-total", type=int, default=500)
    ap.add_argument("--solver-cus", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layout", choices=["range", "xcd-major", "interleaved"], default="xcd-major",
                    help="which CUs the solver gets: the first ones; the first ones of each of the 8 "
                         "XCDs if CU numbers run XCD by XCD; or if they run round the XCDs")
    args = ap.parse_args()
    from bayespy_amd import _lib, device as dev
    L = _lib.lib()
    K, N, M = args.k, args.rows, args.max_total
    rng = np.random.default_rng(0)
    alpha = 0.01 + rng.gamma(2.0, 0.05, size=K)
    totals = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(20260003))
    X = dev.synth_dirmult(N, K, alpha, totals, seed=20260003)
    torch.cuda.synchronize()
    cus = L.dme_device_cus()
    sc = args.solver_cus

    def plain():
        pend = []
        for _ in range(args.steps):
            st = dev.DeviceStats(K, M, M)
            st.add_dense(X, max_count=M, max_total=M)
            pend.append(st.estimate_async(50))
        return [p.result() for p in pend]

    per = cus // 8
    if args.layout == "range":
        solver = set(range(sc))
    elif args.layout == "xcd-major":   # CU i lies on XCD i // 32
        solver = {x * per + j for x in range(8) for j in range(sc // 8)}
    else:                              # CU i lies on XCD i % 8: a range is spread already
        solver = set(range(sc))
    nwords = (cus + 31) // 32

    def words(cu_set):
        w = (ctypes.c_uint32 * nwords)()
        for c in cu_set:
            w[c >> 5] |= 1 << (c & 31)
        return w

    ha, hb = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(L.dme_stream_create_masked(words(set(range(cus)) - solver), nwords, ctypes.byref(ha)))
    _lib.check(L.dme_stream_create_masked(words(solver), nwords, ctypes.byref(hb)))
    A = torch.cuda.ExternalStream(ha.value)
    B = torch.cuda.ExternalStream(hb.value)

    def overlapped():
        pend = []
        for _ in range(args.steps):
            with torch.cuda.stream(A):
                _lib.check(L.dme_set_cu_limit(cus - sc))
                st = dev.DeviceStats(K, M, M)
                st.add_dense(X, max_count=M, max_total=M)
                uv = st.finalize()
                ev = torch.cuda.Event()
                ev.record(A)
            with torch.cuda.stream(B):
                B.wait_event(ev)
                _lib.check(L.dme_set_cu_limit(sc))
                pend.append(dev.PendingEstimate(st, uv, 50, 0.0))
        _lib.check(L.dme_set_cu_limit(0))
        return [p.result() for p in pend]

    def timed(fn):
        fn()  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3, outs

    t_plain, o_plain = timed(plain)
    t_over, o_over = timed(overlapped)
    same = all(np.allclose(a["alpha"], b["alpha"], rtol=1e-9, atol=0) and a["n_iters"] == b["n_iters"]
               for a, b in zip(o_plain, o_over))
    rows_ok = all(o["info"]["rows"] == N for o in o_over)
    print(json.dumps({"rows": N, "K": K, "cus": cus, "solver_cus": sc, "layout": args.layout, "steps": args.steps,
                      "one_stream_ms_per_step": t_plain, "two_masked_streams_ms_per_step": t_over,
                      "same_alphas_and_iterations": same, "rows_counted": rows_ok,
                      "iterations": o_over[-1]["n_iters"]}))
    _lib.check(L.dme_stream_destroy(ha))
    _lib.check(L.dme_stream_destroy(hb))


if __name__ == "__main__":
    main()
