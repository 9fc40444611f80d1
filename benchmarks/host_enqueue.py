import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bayespy_amd import device as dev
K, N, M = 1000, 2_000_000, 500
rng = np.random.default_rng(0)
alpha = 0.01 + rng.gamma(2.0, 0.05, size=K)
tot = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
X = dev.synth_dirmult(N, K, alpha, tot, seed=3)
def step():
    st = dev.DeviceStats(K, M, M)
    st.add_dense(X, max_count=M, max_total=M)
    st.allreduce(same_extents=True, rows_bound=N)
    return st.estimate_async(50, 0.0)
for _ in range(3): step().result()
torch.cuda.synchronize()
import cProfile, pstats
ts = []
pend = []
for i in range(20):
    t0 = time.perf_counter(); pend.append(step()); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
for p in pend: p.result()
print("host enqueue per step: median %.3f ms, min %.3f, max %.3f" % (1e3*np.median(ts), 1e3*min(ts), 1e3*max(ts)))
pr = cProfile.Profile(); pr.enable()
pend = [step() for _ in range(20)]
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
