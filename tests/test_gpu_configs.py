"""BASELINE.json configs 2, 4 and 5 at sizes the oracle finishes in seconds, end to end on the GPU
(generator -> statistics -> scan -> init -> solver) against the CPU oracle on the same matrix.
Integer statistics bit-exact; alphas within 1e-9 relative where the branch history is the same."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _solve_both(oracle, st, ref_st, iterations=50):
    uv = st.finalize()
    U, V = uv.to_lists()
    assert U == [list(map(int, u)) for u in ref_st.U]
    assert V == list(map(int, ref_st.V))
    a0 = st.cli_init_alphas()
    np.testing.assert_allclose(a0.cpu().numpy(), ref_st.cli_init_alphas(), rtol=1e-12)
    out = uv.solve(a0, iterations, 0.0)
    ref = oracle.find_priors(ref_st, ref_st.cli_init_alphas(), iterations, 0.0)
    return out, ref


def test_config2_flipcoins_1m(oracle):
    """C2: K=2 Beta-binomial, alpha=(0.7, 1.2), 60 flips, 1M rows (int32 dense)."""
    from bayespy_amd import device
    N, K = 1_000_000, 2
    alpha_true = np.array([0.7, 1.2])
    totals = torch.full((N,), 60, dtype=torch.int32, device="cuda")
    X = device.synth_dirmult(N, K, alpha_true, totals, seed=20260001, dtype=torch.int32)
    st = device.DeviceStats(K)
    st.add_dense(X, max_count=60, max_total=60)
    assert st.check()["rows"] == N
    ref_st = oracle.append_rows(X.cpu().numpy())
    out, ref = _solve_both(oracle, st, ref_st)
    assert out["exit"] == 1
    # statistical check of the estimate: log-space L2 error (DE/evalDirichletMLE.py:28-34)
    err = math.sqrt(sum((math.log(a) - math.log(t)) ** 2 for a, t in zip(out["alpha"], alpha_true)))
    assert err < 0.02
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
    # mean of the urn sampler: E[x_0] = 60 * 0.7 / 1.9
    assert abs(float(X[:, 0].double().mean()) - 60 * 0.7 / 1.9) < 0.1


def test_config5_heavy_tail(oracle):
    """C5: K=100, row totals min(floor(5*(Pareto(1.2)+1)), 1e5), int32 dense; stresses lenV=1e5,
    ragged U, the global-atomic spill path and multi-segment columns in the solver."""
    from bayespy_amd import device
    N, K = 1_000_000, 100
    rng = np.random.default_rng(20260005)
    alpha_true = 0.05 + rng.gamma(2.0, 0.3, size=K)
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.rand(N, device="cuda", dtype=torch.float64, generator=g)
    totals = torch.clamp(torch.floor(5.0 * (1.0 - u) ** (-1.0 / 1.2)), max=100000).to(torch.int32)
    totals[:3] = torch.tensor([100000, 99999, 65536], dtype=torch.int32)  # the extremes are present
    X = device.synth_dirmult(N, K, alpha_true, totals, seed=20260005, dtype=torch.int32)
    assert torch.equal(X.sum(dim=1).to(torch.int32), totals)
    st = device.DeviceStats(K)
    st.add_dense(X)  # bounds measured by the wrapper
    info = st.check()
    assert info["rows"] == N and info["max_total"] == 100000
    ref_st = oracle.append_rows(X.cpu().numpy())
    assert len(ref_st.V) == 100000
    out, ref = _solve_both(oracle, st, ref_st)
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
    assert out["exit"] == {"A": 1, "B": 2, "C": 3}[ref["exit"]]
    assert out["n_iters"] == ref["n_iters"]


def test_config4_bag_of_words_csr(oracle):
    """C4 shape at small N: K=50k Zipf vocabulary, log-normal document lengths, CSR input."""
    from bayespy_amd import device
    N, K = 20_000, 50_000
    w = 1.0 / np.arange(1, K + 1) ** 1.07
    alpha_true = 50.0 * w / w.sum()
    g = torch.Generator(device="cuda").manual_seed(4)
    ln = torch.exp(math.log(120.0) + 0.6 * torch.randn(N, device="cuda", generator=g))
    totals = torch.clamp(ln, 1, 2000).to(torch.int32)
    X = device.synth_dirmult(N, K, alpha_true, totals, seed=20260004, dtype=torch.uint16)
    Xi = X.view(torch.int16).to(torch.int32)
    del X
    # every word must occur (the reference divides by zero on an empty column, DE/dme.py:69)
    missing = torch.nonzero(Xi.sum(dim=0) == 0).flatten()
    nz = torch.nonzero(Xi)
    rows = torch.cat([nz[:, 0], N + torch.arange(missing.numel(), device="cuda")])
    cols = torch.cat([nz[:, 1], missing]).to(torch.int32)
    vals = torch.cat([Xi[nz[:, 0], nz[:, 1]], torch.ones_like(missing, dtype=torch.int32)])
    Ntot = N + missing.numel()
    rowptr = torch.zeros(Ntot + 1, dtype=torch.int64, device="cuda")
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=Ntot), 0)
    del Xi, nz
    st = device.DeviceStats(K)
    st.add_csr(rowptr, cols.contiguous(), vals.contiguous())
    assert st.check()["rows"] == Ntot
    ref_st = oracle.append_rows_csr(rowptr.cpu().numpy(), cols.cpu().numpy(), vals.cpu().numpy(), K)
    uv = st.finalize()
    U, V = uv.to_lists()
    assert U == [list(map(int, x)) for x in ref_st.U]
    assert V == list(map(int, ref_st.V))
    # CSR mean accumulator uses fp64 atomics: order-dependent in the last bits
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref_st.init_acc, rtol=1e-11)
    a0 = st.cli_init_alphas()
    out = uv.solve(a0, 50, 0.0)
    ref = oracle.find_priors(ref_st, a0.cpu().numpy(), 50, 0.0)
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
    assert out["n_iters"] == ref["n_iters"]


def test_regression_callers_start_point_at_scale(oracle):
    """DirichletRegression/dirichletRegression.py:137-141 takes its baseline prior
    from findDirichletPriors(data, [1.0] * K, 50) - a start far from the optimum, which walks the
    fallback branches (log-space trials, the x0.9 loop) before Newton steps take over.  The drop-in
    module on 1 M rows of K = 200 against the oracle on the same U/V: same alphas to 1e-9, same
    exit, and - when no decision was taken on rounding noise - the same iteration and evaluation
    counts and the same logged loss at every iteration."""
    import bayespy_amd.dirichletMultinomialEstimation as DME
    from bayespy_amd import device
    K, N = 200, 1_000_000
    rng = np.random.default_rng(5)
    alpha_true = 0.02 + rng.gamma(2.0, 0.1, size=K)
    totals = torch.randint(1, 301, (N,), dtype=torch.int32, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(5))
    X = device.synth_dirmult(N, K, alpha_true, totals, seed=20260006)
    data = DME.CompressedRowData(K)
    data.appendDevice(X, max_count=300, max_total=300)
    priors = DME.findDirichletPriors(data, [1.0] * K, 50)
    st = data.deviceStats()
    uv = st.finalize()
    out = uv.solve(torch.ones(K, dtype=torch.float64, device="cuda"), 50, 0.0)
    np.testing.assert_array_equal(np.asarray(priors), out["alpha"])  # the module is this call
    Ul, Vl = uv.to_lists()
    full = oracle.stats_from_lists(Ul, Vl)
    ref = oracle.find_priors(full, np.ones(K), 50, 0.0)
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
    assert {1: "A", 2: "B", 3: "C"}[out["exit"]] == ref["exit"]
    assert out["n_evals"] > out["n_iters"]  # the fallback branches were walked
    if out["fragile_decisions"] == 0:
        assert out["n_iters"] == ref["n_iters"] and out["n_evals"] == ref["n_evals"]
        # the loss and the squared gradient the reference logs at the top of every iteration
        np.testing.assert_allclose(out["trace"][:, 0], ref["trace_loss"], rtol=1e-12)
        # (at the optimum the gradient is what cancellation leaves of terms ~1e6: absolute bound)
        np.testing.assert_allclose(out["trace"][:, 1], ref["trace_gradsq"], rtol=1e-6, atol=1e-11)
