#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

metric : count-rows/sec to converged Dirichlet MLE (K=1000, 10M rows) at 1/2/4/8 GPUs, whole job
         (BASELINE.json's metric string, verbatim)
workload: configs[2] = "K=1000 synthetic Dirichlet-multinomial, 10M rows, row totals 1-500"
          (SURVEY 8d C3): uint16 dense rows resident in HBM, alpha_true = 0.01 + Gamma(2, 0.05)
          (numpy seed 0), n_i ~ UniformInt[1, 500], x_i ~ DirichletMultinomial(n_i, alpha_true),
          generated on the device (dme_synth_dirmult, seed 20260003, keyed on the GLOBAL row index,
          so every world size counts the same matrix).
step    : one full pass of the hot path over the resident matrix:
          statistics kernel -> [RCCL all-reduce of the integer histograms when N > 1] -> suffix
          scan -> CLI initial alphas -> on-device Newton loop to convergence -> status check.
scaling : strong (default) — the metric's 10M rows in TOTAL, rank r holds the contiguous block
          parallel.shard_rows(rows, r, world) (SURVEY 8d C3 "Sharded: N/P contiguous row blocks");
          weak (--scaling weak) — every rank holds its own --rows rows.
          value = all rows of the job / max-over-ranks time per step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--scaling strong|weak] [--dtype u16|i32]
  N > 1: the driver launches one rank per GPU with torch.distributed.run (backend nccl = RCCL);
  run by hand without it, `python bench.py --gpus N` starts those ranks itself (a child
  torch.distributed.run, before this process touches the GPU) and exits with its status.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the statistics
launch of every timed step; `phases_ms` (stats / all-reduce / scan + init + solve) comes from a few
extra, untimed steps with more events; `cpu_baseline` times the CPU oracle (oracle/: a C port, a
CPython-loop port and a NumPy-structured port of the reference) on bounded row samples on this box's
host cores (rank 0, N = 1 only) and checks the GPU's alphas against the oracle's (exit status 1 if
they differ by more than 1e-9 relative).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALPHA_RTOL = 1e-9      # BASELINE.json north_star: alphas within 1e-9 relative of the reference


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000,
                    help="rows of the whole job (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--max-total", type=int, default=500)
    ap.add_argument("--dtype", choices=["u16", "i32"], default="u16")
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-port-rows", type=int, default=20_000,
                    help="rows the CPython-loop port is timed on (SURVEY 8d: 20 000 rows x 3, median)")
    ap.add_argument("--cpu-numpy-port-rows", type=int, default=5_000,
                    help="rows the NumPy-structured port is timed on, x 3, median (it runs at ~1.5e3 "
                         "rows/s: 20 000 rows x 3 would add 40 s to every bench run)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals "
                         "with several ranks on one GPU)")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed statistics passes before the warm-up steps (see main)")
    ap.add_argument("--phase-steps", type=int, default=5,
                    help="extra untimed steps that carry the per-phase events")
    ap.add_argument("--sync-steps", action="store_true",
                    help="wait on the host after every timed step instead of enqueueing the steps "
                         "back to back (the headline then equals `sync_ms_per_step`)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: create the process group anyway and send the all-reduce "
                         "message through it (rehearses the RCCL code path on a single GPU)")
    ap.add_argument("--verify", action="store_true",
                    help="check size-independent properties of the statistics after the run")
    return ap.parse_args()


def git_blob_sha1(path):
    """The git blob id of a file (sha1 of "blob <size>\\0" + content), computed without git."""
    import hashlib
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def median_time(fn, repeats):
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def launch_ranks(args):
    """--gpus N without a torch.distributed.run parent: start the N ranks as a child job (this
    process has not touched the GPU) and exit with its status.  Never continues as one GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching %s\n" % (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd))


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: with NCCL_DEBUG=VERSION (set on the GPU
    boxes) RCCL prints its banner to stdout from C when the communicator is created, and stdout is
    for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    if os.environ.get("DME_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DME_BENCH_WATCHDOG"]), exit=True)
    import torch
    import torch.distributed as dist

    from bayespy_amd import device as dev
    from bayespy_amd.parallel import shard_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {ndev} visible")
    local_rank %= ndev  # gloo rehearsals: several ranks share a card
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    forced = world == 1 and args.force_collective
    if forced:
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                # the communicator is bound to this rank's GPU before any collective runs
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(args.backend)
            # the communicator itself is created by the first collective
            dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=device))
            torch.cuda.synchronize()

    K, M = args.k, args.max_total
    if args.scaling == "strong":
        total_rows = args.rows
        lo, hi = shard_rows(total_rows, rank, world)
    else:
        total_rows = args.rows * world
        lo, hi = rank * args.rows, (rank + 1) * args.rows
    N = hi - lo  # this rank's rows
    if N <= 0:
        raise SystemExit("bench.py: fewer rows than ranks")
    tdtype = torch.uint16 if args.dtype == "u16" else torch.int32
    esize = 2 if args.dtype == "u16" else 4

    # ---- synthetic workload, generated on the device, not timed --------------
    # Row totals and rows are functions of the GLOBAL row index: the job's matrix is the same for
    # every world size, a rank generates its own block of it.
    rng = np.random.default_rng(0)
    alpha_true = 0.01 + rng.gamma(2.0, 0.05, size=K)
    t0 = time.time()
    gen = torch.Generator(device=device).manual_seed(20260003)
    totals_all = torch.randint(1, M + 1, (total_rows,), dtype=torch.int32, device=device, generator=gen)
    totals = totals_all[lo:hi].contiguous()
    del totals_all
    X = dev.synth_dirmult(N, K, alpha_true, totals, seed=20260003, dtype=tdtype,
                          row_offset=lo, device=device)
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def one_step(timers=None, phases=None, wait=True):
        """One pass of the hot path.  wait=False returns the enqueued job (a PendingEstimate): the
        host goes on to enqueue the next step while the GPU works, and the results are collected -
        and checked - before the clock stops."""
        st = dev.DeviceStats(K, M, M, device=device)
        if timers is not None or phases is not None:
            e0, e1 = ev(), ev()
            e0.record()
        st.add_dense(X, max_count=M, max_total=M)
        if timers is not None or phases is not None:
            e1.record()
        # every rank sized its histograms for the same M; every bin count is at most total_rows
        st.allreduce(same_extents=True, rows_bound=total_rows, force_message=forced)
        if phases is not None:
            e2 = ev()
            e2.record()
        pend = st.estimate_async(args.iterations, 0.0)  # scan + CLI initial alphas + Newton loop
        if timers is not None:
            timers.append((e0, e1))
        if not wait:
            return pend
        out = pend.result()  # synchronises; raises on data errors / solver errors
        if phases is not None:
            e3 = ev()
            e3.record()
            e3.synchronize()
            phases.append((e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)))
        return out

    # initialisation that is not a step: code objects are loaded and, with several ranks, the RCCL
    # communicator is created (a tiny collective), so that even --warmup 0 times steady-state steps
    prime = dev.DeviceStats(K, M, M, device=device)
    prime.add_dense(X[:4096], max_count=M, max_total=M)
    if world > 1 or forced:
        dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=device))
    prime.check()
    del prime
    # Settle (untimed, not a step): for up to a second after ANOTHER process has released tens of GB
    # of device memory (a test run before this one), the same statistics launch on the same data
    # takes 3.44 ms instead of 3.32 ms, then drops back by itself (profiles/r1_notes.md: measured
    # as a function of time since that process ended, whatever the number of steps run).  The passes
    # below only let that time go by with the device busy, so that the timed steps measure the
    # steady state a job of more than a second sees.
    t_settle = time.time()
    n_settle = 0
    while time.time() - t_settle < args.settle_seconds:
        tmp = dev.DeviceStats(K, M, M, device=device)
        tmp.add_dense(X, max_count=M, max_total=M)
        tmp.check()
        n_settle += 1

    for _ in range(args.warmup):
        one_step()

    def barrier():
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(pipelined):
        timers, outs = [], []
        barrier()
        t_start = time.perf_counter()
        if pipelined:
            pends = [one_step(timers, wait=False) for _ in range(args.steps)]
            barrier()
            outs = [p.result() for p in pends]  # host reads of pinned words: every job is checked
        else:
            outs = [one_step(timers) for _ in range(args.steps)]
            barrier()
        elapsed = time.perf_counter() - t_start
        if world > 1 or forced:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.cpu()[0])
        for o in outs:
            # proof that the collective saw every rank: the reduced statistics count the job's rows
            assert o["info"]["rows"] == total_rows, (o["info"], total_rows)
            assert o["exit"] in (1, 2, 3)
        return elapsed, timers, outs

    # the K timed steps: enqueued back to back (no host wait between steps), synchronised and checked
    # before the clock stops; then, outside the headline, the same K steps with a host wait after each
    elapsed, timers, outs = timed(pipelined=not args.sync_steps)
    out = outs[-1]
    uv = out["uv"]
    uv_stats = outs[-1].get("stats")
    sync_elapsed = elapsed if args.sync_steps else timed(pipelined=False)[0]

    ms_per_step = elapsed / args.steps * 1e3
    stats_ms = float(np.mean([a.elapsed_time(b) for a, b in timers]))
    value = total_rows / (elapsed / args.steps)

    # per-phase times from extra steps outside the timed region (an event costs device time)
    phases = []
    for _ in range(max(args.phase_steps, 0)):
        one_step(phases=phases)
    barrier()
    ph = np.mean(np.asarray(phases, dtype=np.float64), axis=0) if phases else None

    algo_bytes = N * K * esize  # this GPU's block of the count matrix, read exactly once
    achieved = algo_bytes / (stats_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes, benchmarks/pmc_bench_traffic.sh -> profiles/traffic.json).  The file names the git blob of
    # the kernel source it was measured on: a tree whose dme_stats.hip differs reports null, not a stale
    # number.
    traffic, traffic_note = None, "profiles/traffic.json absent"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            blob = git_blob_sha1(os.path.join(ROOT, "bayespy_amd", "csrc", "dme_stats.hip"))
            blob_h = git_blob_sha1(os.path.join(ROOT, "bayespy_amd", "csrc", "dme_common.h"))
            if tj.get("dme_stats_hip_blob") != blob:
                traffic_note = ("profiles/traffic.json was measured on dme_stats.hip blob %s, this tree "
                                "holds %s" % (tj.get("dme_stats_hip_blob"), blob))
            elif tj.get("dme_common_h_blob") != blob_h:  # the header the kernels include
                traffic_note = ("profiles/traffic.json was measured on dme_common.h blob %s, this tree "
                                "holds %s" % (tj.get("dme_common_h_blob"), blob_h))
            elif tj.get("N") != N or tj.get("K") != K:
                traffic_note = "profiles/traffic.json is for N=%s K=%s" % (tj.get("N"), tj.get("K"))
            else:  # PMC traffic is per launch of the profiled shape only
                traffic = tj.get(f"stats_dense_{args.dtype}_bytes_per_launch")
                traffic_note = "PMC FETCH_SIZE/WRITE_SIZE, commit %s" % tj.get("commit")
        except Exception as ex:
            traffic, traffic_note = None, "profiles/traffic.json unreadable: %s" % ex

    line = {
        "metric": "count-rows/sec to converged Dirichlet MLE (K=1000, 10M rows) at 1/2/4/8 GPUs",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("K=%d synthetic Dirichlet-multinomial, %d rows in all (%s scaling: %d rows "
                         "on each of %d GPU%s), row totals 1-%d, %s dense (BASELINE configs[2])"
                         % (K, total_rows, args.scaling, N, world, "" if world == 1 else "s", M,
                            "uint16" if esize == 2 else "int32")),
            "K": K, "rows_per_gpu": N, "rows_total": total_rows, "max_row_total": M,
            "input_dtype": args.dtype, "backend": args.backend if (world > 1 or forced) else None,
            "parallelism": "rows sharded x%d (contiguous blocks), one integer all-reduce, "
                           "replicated solve" % world,
        },
        "roofline": {
            "bound": "hbm", "kernel": "stats_dense_kernel", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": stats_ms, "per": "GPU (rank 0)",
        },
        "phases_ms": {
            "stats": stats_ms,
            "rest (all-reduce + scan + init + solve + check + host)": ms_per_step - stats_ms,
        },
        "step_issue": ("a host wait after every step" if args.sync_steps else
                       "the K steps are enqueued back to back on one stream; one synchronise, then "
                       "every job's result is checked, before the clock stops"),
        "sync_ms_per_step": sync_elapsed / args.steps * 1e3,
        "solver": {"iterations": out["n_iters"], "loss_evals": out["n_evals"], "exit": out["exit"],
                   "launches": out["launches"], "final_loss": out["final_loss"],
                   "fragile_decisions": out["fragile_decisions"]},
        # an accept / reject decision of the Newton loop was taken on a loss difference of
        # a few ulp: the reference's sequential sums may decide it the other way
        "parity_fragile": out["fragile_decisions"] > 0,
        "gen_seconds": t_gen,
        "settle": {"seconds": args.settle_seconds, "untimed_statistics_passes": n_settle},
    }
    if ph is not None:
        line["phases_ms"]["untimed_steps_with_events"] = {
            "steps": len(phases), "stats": float(ph[0]), "allreduce": float(ph[1]),
            "scan+init+solve+check": float(ph[2])}

    if args.verify and rank == 0:
        U = uv.Ui.view(K, uv.ldU)
        assert bool((U[:, 1:] <= U[:, :-1]).all())
        assert int(uv.Vi[0]) == total_rows  # every row has total >= 1
        if world == 1:
            assert int(uv.Vi.sum()) == int(totals.to(torch.int64).sum())
        line["verified"] = True

    # ---- CPU baseline: the oracle (ports of the reference loops) on bounded samples ---------------
    parity_ok = True
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        ns = min(args.cpu_sample_rows, N)
        Xh = X[:ns].cpu().numpy()
        t_stats_cpu, all_c = median_time(lambda: oracle.append_rows(Xh, capU=M, capV=M),
                                         max(args.cpu_repeats, 1))
        # the solver's cost does not depend on N: time it on the full-size statistics
        Ufull, Vfull = uv.to_lists()
        full = oracle.stats_from_lists(Ufull, Vfull)
        a0 = uv_stats.cli_init_alphas().cpu().numpy()
        t0 = time.perf_counter()
        sol = oracle.find_priors(full, a0, args.iterations, 0.0)
        t_solve_cpu = time.perf_counter() - t0
        cpu_total = t_stats_cpu * (N / ns) + t_solve_cpu
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        line["cpu_baseline"] = {
            "value": N / cpu_total, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": ("oracle/dme_oracle.c (single-threaded C port of appendRow + mean accumulator) "
                       "timed on the first %d rows x 3 runs: median %.2f s, EXTRAPOLATED linearly to %d rows; plus the "
                       "oracle solver on the full-size U/V: %.3f s (%d iterations, %d loss evals)"
                       % (ns, t_stats_cpu, N, t_solve_cpu, sol["n_iters"], sol["n_evals"])),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
        }
        # the reference's own cost models — interpreted loops per row (DE) and ndarray slices grown by
        # np.append per column per row (NumPy port) — on smaller samples, statistics build only
        from oracle import np_port, py_port
        reps = max(args.cpu_repeats, 1)
        npy = min(args.cpu_port_rows, ns)
        rows_py = Xh[:npy].astype(np.int64).tolist()
        t_py, all_py = median_time(lambda: py_port.build(rows_py, K), reps)
        line["cpu_baseline"]["cpython_port"] = {
            "value": npy / t_py, "unit": "rows/s", "cores": 1,
            "sample": "oracle/py_port.py (CPython loops of DE/dme.py:113-123 + DE/fdp.py:87-89) on "
                      "%d rows x %d runs, median %.2f s (%s); statistics build only; rows/s holds for "
                      "any N (O(N))" % (npy, reps, t_py, ", ".join("%.2f" % t for t in all_py))}
        nnp = min(args.cpu_numpy_port_rows, ns)
        rows_np = rows_py[:nnp]
        t_np, all_np = median_time(lambda: np_port.build(rows_np, K), reps)
        line["cpu_baseline"]["numpy_port"] = {
            "value": nnp / t_np, "unit": "rows/s", "cores": 1,
            "sample": "oracle/np_port.py (ndarray U/V grown by np.append, a slice add per column per "
                      "row: NumpyPort/dirichletMultinomialEstimation.py:106-121, + DE/fdp.py:87-89) on "
                      "%d rows x %d runs, median %.2f s (%s); statistics build only; rows/s holds for "
                      "any N (O(N))" % (nnp, reps, t_np, ", ".join("%.2f" % t for t in all_np))}
        rel = float(np.max(np.abs(out["alpha"] - sol["alpha"]) / np.abs(sol["alpha"])))
        ref_loss = oracle.total_loss(sol["alpha"], full, 0.0)
        rel_loss = abs(out["final_loss"] - ref_loss) / abs(ref_loss)
        line["cpu_baseline"]["gpu_vs_oracle_max_rel_alpha_err"] = rel
        line["cpu_baseline"]["gpu_vs_oracle_rel_loss_err"] = rel_loss
        line["cpu_baseline"]["gpu_vs_oracle_same_counts"] = (
            out["n_iters"] == sol["n_iters"] and out["n_evals"] == sol["n_evals"])
        parity_ok = rel <= ALPHA_RTOL

    if rank == 0:
        print(json.dumps(line))
    if world > 1 or forced:
        dist.destroy_process_group()
    if not parity_ok:
        sys.stderr.write("bench.py: GPU alphas differ from the oracle's by %.3e relative (> %g)\n"
                         % (line["cpu_baseline"]["gpu_vs_oracle_max_rel_alpha_err"], ALPHA_RTOL))
        raise SystemExit(1)


if __name__ == "__main__":
    main()
