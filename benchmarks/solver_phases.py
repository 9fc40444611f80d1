#!/usr/bin/env python3
"""Where one launch of the solver (iter_kernel, K <= 2048) spends its time.

Needs a build of the library with time stamps compiled in (never the shipped one):
    make -C bayespy_amd/csrc ts          # -> bayespy_amd/libdme_ts.so  (-DDME_SOLVE_TS)
    python benchmarks/solver_phases.py
Thread 0 of workgroup 0 reads s_memrealtime (10 ns) at the phase boundaries of every launch and
leaves the differences in unused rows of the trace buffer; printed per launch: state and partial
sums loaded, step_body, state stored by workgroup 0, evaluation of the new trial point; and the
pieces of step_body for launch 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
from bayespy_amd import _lib
_lib.LIB_PATH = os.path.abspath("bayespy_amd/libdme_ts.so")
from bayespy_amd import device as dev
if os.environ.get("HEAVY"):  # BASELINE configs[4]: K = 100, heavy-tailed row totals (benchmarks/heavy_tail.py)
    K, N = 100, 1_000_000
    rng = np.random.default_rng(20260005)
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    u = torch.rand(N, device="cuda", dtype=torch.float64, generator=torch.Generator(device="cuda").manual_seed(5))
    tot = torch.clamp(torch.floor(5.0 * (1.0 - u) ** (-1.0 / 1.2)), max=100000).to(torch.int32)
    X = dev.synth_dirmult(N, K, alpha, tot, seed=20260005, dtype=torch.int32)
    M = int(tot.max())
else:
    K, N, M = 1000, 2_000_000, 500
    rng = np.random.default_rng(0)
    alpha = 0.01 + rng.gamma(2.0, 0.05, size=K)
    tot = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    X = dev.synth_dirmult(N, K, alpha, tot, seed=3)
for rep in range(3):
    st = dev.DeviceStats(K, M, M)
    st.add_dense(X, max_count=M, max_total=M)
    p = st.estimate_async(50)
    out = p.result()
    tr = p.buf[16:].view(p.cap, 4)[22:22 + 14].cpu().numpy() * 10.0  # ns
    fine = p.buf[16:].view(p.cap, 4)[40:43].cpu().numpy().reshape(-1)[:9] * 10.0
print("iters", out["n_iters"], "evals", out["n_evals"])
print("step_body of launch 5 (ns): gather_eval, loss_from, accept copy, trace+checks, trial, finish_trial | inside gather_eval: V sums, columns, block reduction:", fine)
print("launch: load+sync  step_body  writer_store  eval   (ns)")
for i, r in enumerate(tr):
    print(i, r)
