#!/usr/bin/env python3
"""BASELINE configs[3]: K = 50 000 bag-of-words rows in CSR, one GPU's shard.

Input generator (torch, on the device; benchmark input, not the product path) + a timing loop
around the CSR statistics entry (dme_stats_csr_i32) and the rest of the pipeline.

  doc length n_i ~ clip(LogNormal(ln 120, 0.6), 1, 2000)
  words      ~ Dirichlet-multinomial(n_i, alpha),  alpha_k = 50 * Zipf(1.07) weights (sum 50)
               drawn as a Polya urn: token t is a fresh draw from alpha/50 with probability
               50/(50+t), else a copy of a uniformly chosen earlier token of the same row
  every column non-empty: one extra single-word row per missing word
  seed 20260004 (+ rank);  colidx int32, vals int32, rowptr int64

  python benchmarks/bow_csr.py [--rows 1000000] [--k 50000] [--steps 5] [--solve]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      benchmarks/bow_csr.py --rows 12500000 --solve      # the full configuration: 100M rows on 8 GPUs
Prints one JSON line (rank 0): nnz, algorithmic bytes (nnz*8 + (N+1)*8), ms and GB/s of the
statistics pass of one rank; with several ranks also the time of the whole estimation (statistics,
all-reduce, scan, solve; max over ranks).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def zipf_alpha(K, s=1.07, total=50.0):
    import numpy as np
    w = np.arange(1, K + 1, dtype=np.float64) ** (-s)
    return total * w / w.sum()


def generate(N, K, seed=20260004, device="cuda", chunk=500_000, s=1.07, total=50.0,
             shuffle_columns=True):
    """Returns (rowptr int64[N'+1], colidx int32[nnz], vals int32[nnz]) on the device, N' >= N
    (rows appended for words nobody drew).  shuffle_columns: word frequency rank is not the
    column index (as in a real vocabulary)."""
    import numpy as np
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    alpha = zipf_alpha(K, s, total)
    cdf = torch.from_numpy(np.cumsum(alpha) / alpha.sum()).to(dev)
    perm = None
    if shuffle_columns:
        perm = torch.from_numpy(np.random.default_rng(seed).permutation(K)).to(dev)
    cols, vals, lens = [], [], []
    for lo in range(0, N, chunk):
        n_rows = min(chunk, N - lo)
        ln = torch.empty(n_rows, dtype=torch.float64, device=dev).normal_(math.log(120.0), 0.6, generator=g)
        n = ln.exp().floor().clamp_(1, 2000).to(torch.int64)
        start = torch.cumsum(n, 0) - n
        T = int(n.sum())
        row = torch.repeat_interleave(torch.arange(n_rows, device=dev), n, output_size=T)
        t = torch.arange(T, device=dev) - start[row]
        u = torch.rand(T, dtype=torch.float64, device=dev, generator=g)
        fresh = u * (total + t.to(torch.float64)) < total
        u2 = torch.rand(T, dtype=torch.float64, device=dev, generator=g)
        parent = start[row] + (u2 * t.to(torch.float64)).floor().to(torch.int64)
        ptr = torch.where(fresh, torch.arange(T, device=dev), parent)
        del parent, u2
        for _ in range(12):  # pointer jumping: chains are at most 2000 long
            ptr = ptr[ptr]
        u3 = torch.rand(T, dtype=torch.float64, device=dev, generator=g)
        word = torch.searchsorted(cdf, u3).clamp_(max=K - 1)
        word = word[ptr]
        if perm is not None:
            word = perm[word]
        key = row * K + word
        del row, t, u, u3, ptr, word, fresh
        key, _ = torch.sort(key)
        uk, cnt = torch.unique_consecutive(key, return_counts=True)
        del key
        r = torch.div(uk, K, rounding_mode="floor")
        cols.append((uk - r * K).to(torch.int32))
        vals.append(cnt.to(torch.int32))
        lens.append(torch.bincount(r, minlength=n_rows))
        del uk, cnt, r
    colidx = torch.cat(cols)
    v = torch.cat(vals)
    rl = torch.cat(lens)
    seen = torch.zeros(K, dtype=torch.bool, device=dev)
    seen[colidx.to(torch.int64)] = True
    missing = torch.nonzero(~seen).flatten().to(torch.int32)
    if missing.numel():
        colidx = torch.cat([colidx, missing])
        v = torch.cat([v, torch.ones_like(missing)])
        rl = torch.cat([rl, torch.ones(missing.numel(), dtype=rl.dtype, device=dev)])
    rowptr = torch.zeros(rl.numel() + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(rl, 0)
    return rowptr, colidx.contiguous(), v.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=50_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--solve", action="store_true")
    ap.add_argument("--no-shuffle", action="store_true")
    ap.add_argument("--zipf-s", type=float, default=1.07,
                    help="exponent of the word weights (1.07 = BASELINE configs[3]; 0 = no frequent words: an experiment)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--dense-message", action="store_true",
                    help="all-reduce K x nbins words instead of the occupied bins of every column")
    ap.add_argument("--lib", default=None, help="another build of libdme_hip.so to time (A/B comparisons)")
    ap.add_argument("--free-first", action="store_true",
                    help="drop the previous pass's DeviceStats (histograms, workspace) before the next one is made: "
                         "every pass then gets the same blocks from torch's allocator instead of two sets in turn")
    args = ap.parse_args()
    if args.lib:
        from bayespy_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    import torch
    import torch.distributed as dist
    from bayespy_amd import device as dev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    t0 = time.time()
    rowptr, colidx, vals = generate(args.rows, args.k, seed=20260004 + rank, s=args.zipf_s,
                                    shuffle_columns=not args.no_shuffle)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    N = rowptr.numel() - 1
    nnz = colidx.numel()
    max_count = int(vals.max())
    cs = torch.zeros(nnz + 1, dtype=torch.int64, device=vals.device)
    cs[1:] = torch.cumsum(vals.to(torch.int64), 0)
    max_total = int((cs[rowptr[1:]] - cs[rowptr[:-1]]).max())
    del cs
    if world > 1:  # every rank sizes its histograms alike: one collective per estimation
        b = torch.tensor([max_count, max_total], dtype=torch.int64, device=vals.device)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        max_count, max_total = (int(x) for x in b.cpu())
    algo = nnz * 8 + (N + 1) * 8

    def one():
        st = dev.DeviceStats(args.k, max_count, max_total)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.add_csr(rowptr, colidx, vals, max_count=max_count, max_total=max_total)
        e1.record()
        return st, e0, e1

    for _ in range(args.warmup):
        one()
    ms = []
    st = None
    for _ in range(args.steps):
        if args.free_first:
            st = None
        st, e0, e1 = one()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    info = st.check()
    if world > 1:
        # the whole estimation, all ranks: statistics -> all-reduce -> scan -> replicated solve
        whole = []
        for _ in range(args.steps):
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stw = dev.DeviceStats(args.k, max_count, max_total)
            stw.add_csr(rowptr, colidx, vals, max_count=max_count, max_total=max_total)
            msg_info = {}
            te0 = time.perf_counter()
            stw.allreduce(same_extents=True, rows_bound=world * (1 << 28),
                          compact=not args.dense_message, info=msg_info)
            torch.cuda.synchronize()
            t_reduce = time.perf_counter() - te0
            uvw = stw.finalize()
            outw = uvw.solve(stw.cli_init_alphas(), 50, 0.0, trace_alphas=False)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=vals.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            whole.append(float(t.cpu()[0]))
        rows_all = torch.tensor([N], dtype=torch.int64, device=vals.device)
        dist.all_reduce(rows_all)
    line = {"workload": "bag-of-words CSR (BASELINE configs[3], one GPU's shard)", "K": args.k,
            "rows": N, "nnz": nnz, "max_count": max_count, "max_total": max_total,
            "algorithmic_bytes": algo, "stats_ms": sum(ms) / len(ms), "stats_ms_min": min(ms),
            "achieved_GBps": algo / (sum(ms) / len(ms) * 1e-3) / 1e9, "gen_seconds": t_gen,
            "stats_ms_all": [round(x, 4) for x in ms],
            "rows_checked": info["rows"]}
    if args.solve:
        t0 = time.perf_counter()
        uv = st.finalize()
        a0 = st.cli_init_alphas()
        out = uv.solve(a0, 50, 0.0, trace_alphas=False)
        line["finalize_solve_ms"] = (time.perf_counter() - t0) * 1e3
        line["solver"] = {"iterations": out["n_iters"], "loss_evals": out["n_evals"],
                          "exit": out["exit"], "final_loss": out["final_loss"]}
    if world > 1:
        line["n_gpus"] = world
        line["rows_all_ranks"] = int(rows_all.cpu()[0])
        line["estimation_ms_all_ranks"] = min(whole) * 1e3
        line["rows_per_s_all_ranks"] = int(rows_all.cpu()[0]) / min(whole)
        line["rows_checked_all_ranks"] = stw.check()["rows"]
        line["allreduce"] = {"message_bytes": msg_info["message_bytes"],
                             "extents_bytes": msg_info.get("extents_bytes", 0),
                             "dense_message_bytes": 4 * (args.k * stw.nbins + stw.tbins),
                             "form": "dense" if args.dense_message else "compact",
                             "ms_incl_pack_unpack_last_step": t_reduce * 1e3}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
