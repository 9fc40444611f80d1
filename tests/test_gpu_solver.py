"""GPU parity of the solver (dme_eval, dme_predicted_step*, dme_solve) against golden vectors
produced by the reference (DE variant) and against the CPU oracle.

Tolerances (BASELINE.json north_star): alphas within 1e-9 relative, log-likelihood / loss within
1e-12 relative.  The device sums in a different (parallel, compensated) order than the
reference's sequential loops, so bit equality is not expected for floating point."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

RTOL_ALPHA = 1e-9
RTOL_LOSS = 1e-12


def _uv(fx):
    from bayespy_amd import device
    return device.DeviceUV.from_lists(fx["U"], fx["V"])


@pytest.mark.parametrize("name", [c for c in GOLDEN_CASES])
def test_function_level(name):
    from bayespy_amd import device
    fx = load_golden(name)
    uv = _uv(fx)
    for rec in fx["funcs"]:
        g = rec["DE"]
        lp, hc, grad, hd, gsq = uv.evaluate(g["alpha"], g["hyperprior"])
        np.testing.assert_allclose(lp, g["logprob"], rtol=RTOL_LOSS)
        np.testing.assert_allclose(hc, g["hconst"], rtol=1e-12)
        np.testing.assert_allclose(hd, g["hdiag"], rtol=1e-12)
        # the gradient is a difference of two large sums: absolute tolerance scaled by them
        scale = max(1.0, np.max(np.abs(g["hdiag"])) ** 0.5, abs(g["hconst"]) ** 0.5)
        tsub = np.max(np.abs(g["grad"])) + 1.0
        np.testing.assert_allclose(grad, g["grad"], rtol=1e-9, atol=1e-12 * tsub * scale)
        if g["step"] is None:
            with pytest.raises(ZeroDivisionError):
                device.predicted_step(g["hconst"], g["hdiag"], g["grad"])
        else:
            np.testing.assert_allclose(device.predicted_step(g["hconst"], g["hdiag"], g["grad"]),
                                       g["step"], rtol=1e-11, atol=1e-18)
            np.testing.assert_allclose(
                device.predicted_step(g["hconst"], g["hdiag"], g["grad"], alphas=g["alpha"]),
                g["stepAlt"], rtol=1e-11, atol=1e-18)


def _solve_cases():
    out = []
    for name in GOLDEN_CASES:
        fx = load_golden(name)
        for i, s in enumerate(fx["solves"]):
            if s["variant"] == "DE":
                out.append((name, i))
    return out


# Branch histories that the reference itself does not reproduce between its own DE and NumPy
# variants (the x0.9 halving loop ends on a loss difference below one ulp): only the returned
# alphas / exit are compared there, not the evaluation count.
NOISE_LIMITED = {("dirmult_k50", "cli", 0.01), ("dirmult_k50", "ones", 0.01)}
# A trajectory that diverges (alpha grows to 2.5e6 in 50 iterations, exit C): the reference's own
# DE and NumPy variants differ there by 4.2e-12 in alpha and 4.3e-12 in the loss, so the loss
# tolerance is 1e-10 for it (alphas still meet 1e-9).
DIVERGING = {("test_csv", "fifty", 0.25)}


@pytest.mark.parametrize("name,idx", _solve_cases())
def test_solve_matches_reference(name, idx):
    fx = load_golden(name)
    s = fx["solves"][idx]
    uv = _uv(fx)
    if "raises" in s:
        with pytest.raises(ZeroDivisionError):
            uv.solve(s["init"], s["iterations"], s["hyperprior"])
        return
    out = uv.solve(s["init"], s["iterations"], s["hyperprior"])
    key = (name, s["init_name"], s["hyperprior"])
    exits = {"A": 1, "B": 2, "C": 3}
    assert out["exit"] == exits[s["exit"]], key
    np.testing.assert_allclose(out["alpha"], s["final"], rtol=RTOL_ALPHA, err_msg=str(key))
    rl = 1e-10 if key in DIVERGING else RTOL_LOSS
    np.testing.assert_allclose(out["final_loss"], s["final_loss"], rtol=rl)
    assert out["n_iters"] == s["n_iters"], key
    if key not in NOISE_LIMITED:
        assert out["n_evals"] == s["n_evals"], key
    tl = [t["loss"] for t in s["trace"]]
    np.testing.assert_allclose(out["trace"][:, 0], tl, rtol=rl)
    tg = np.array([t["gradSq"] for t in s["trace"]])
    np.testing.assert_allclose(out["trace"][:, 1], tg, rtol=1e-6, atol=1e-9)
    ta = np.array([t["priors"] for t in s["trace"]])
    np.testing.assert_allclose(out["trace_alpha"], ta, rtol=RTOL_ALPHA)


def test_solve_vs_oracle_on_gpu_built_stats(oracle):
    """End to end on the device: statistics kernel -> scan -> init alphas -> solver, against the
    oracle run on the same matrix."""
    from bayespy_amd import device
    for name in ("dirmult_k1000", "dirmult_k50", "heavy_k20", "flipcoins_1500"):
        fx = load_golden(name)
        X = fx["X"]
        st = device.DeviceStats(fx["K"])
        st.add_dense(torch.from_numpy(np.ascontiguousarray(X.astype(np.int32))).cuda())
        st.check()
        uv = st.finalize()
        a0 = st.cli_init_alphas()
        out = uv.solve(a0, 50, 0.0)
        ref_st = oracle.append_rows(X.astype(np.int32))
        ref = oracle.find_priors(ref_st, ref_st.cli_init_alphas(), 50, 0.0)
        np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=RTOL_ALPHA)
        assert out["n_iters"] == ref["n_iters"] and out["n_evals"] == ref["n_evals"]
        np.testing.assert_allclose(out["final_loss"],
                                   oracle.total_loss(ref["alpha"], ref_st, 0.0), rtol=RTOL_LOSS)


def test_nonpositive_init_raises():
    fx = load_golden("test_csv")
    uv = _uv(fx)
    with pytest.raises(ValueError):
        uv.solve([0.0, 1.0], 50, 0.0)


def test_zero_iterations_returns_init():
    fx = load_golden("test_csv")
    uv = _uv(fx)
    out = uv.solve([0.5, 0.25], 0, 0.0)
    assert out["exit"] == 3 and out["n_iters"] == 0
    assert out["alpha"].tolist() == [0.5, 0.25]


def test_single_launch_and_chained_solvers_agree_bit_for_bit():
    """The three launch structures of the solver - the whole loop in one launch (small problems), one
    launch per evaluation with the step repeated by every workgroup (K <= 2048), an (evaluate, step)
    launch pair per evaluation - run the same device functions in the same order of operations, so
    they return the same bits: checked on every golden solve, all branches and exits included."""
    from bayespy_amd import _lib
    L = _lib.lib()
    CHAIN, SINGLE, FUSED = 1, 2, 3
    n = 0
    try:
        for name in GOLDEN_CASES:
            fx = load_golden(name)
            uv = _uv(fx)
            for s in fx["solves"]:
                if s["variant"] != "DE" or "raises" in s:
                    continue
                outs = []
                for path in (SINGLE, CHAIN, FUSED):
                    if path == SINGLE and fx["K"] > 100:
                        continue  # one workgroup for everything: minutes on the K = 1000 fixture
                    _lib.check(L.dme_solve_set_path(path))
                    outs.append(uv.solve(s["init"], s["iterations"], s["hyperprior"]))
                a = outs[0]
                for b in outs[1:]:
                    assert a["exit"] == b["exit"] and a["n_iters"] == b["n_iters"] and a["n_evals"] == b["n_evals"]
                    assert np.array_equal(a["alpha"], b["alpha"]), (name, s["init_name"])
                    assert np.array_equal(a["trace"], b["trace"])
                    assert np.array_equal(a["trace_alpha"], b["trace_alpha"])
                    assert a["final_loss"] == b["final_loss"]
                    assert a["fragile_decisions"] == b["fragile_decisions"]
                n += 1
    finally:
        _lib.check(L.dme_solve_set_path(0))
    assert n >= 10
    assert L.dme_solve_set_path(9) == _lib.ERR_INVALID_ARG


def test_fused_path_raises_like_the_chain():
    """Errors travel the same way on the one-launch-per-evaluation path: an all-zero column is a
    ZeroDivisionError, a non-positive start a ValueError."""
    from bayespy_amd import _lib
    L = _lib.lib()
    fx = load_golden("zero_column")
    uv = _uv(fx)
    try:
        _lib.check(L.dme_solve_set_path(3))
        for s in fx["solves"]:
            if s["variant"] == "DE" and "raises" in s:
                with pytest.raises(ZeroDivisionError):
                    uv.solve(s["init"], s["iterations"], s["hyperprior"])
        with pytest.raises(ValueError):
            uv.solve([0.0] + [1.0] * (fx["K"] - 1), 50, 0.0)
    finally:
        _lib.check(L.dme_solve_set_path(0))


def test_enqueued_estimate_equals_the_blocking_calls():
    """DeviceStats.estimate_async (dme_solve_enqueue: set-up computes the CLI's initial alphas from
    the mean accumulators, the status words ride in the result block, nobody waits) against the
    blocking sequence finalize / cli_init_alphas / solve / check: the same bits.  Several jobs are
    in flight at once, and one whose default batch of launches falls short continues."""
    from bayespy_amd import device
    fx = load_golden("dirmult_k1000")
    X = torch.from_numpy(fx["X"].astype(np.int32)).cuda()
    pends = []
    for _ in range(3):
        st = device.DeviceStats(fx["K"])
        st.add_dense(X)
        pends.append(st.estimate_async(50, 0.0))
    ref_st = device.DeviceStats(fx["K"])
    ref_st.add_dense(X)
    info = ref_st.check()
    ref = ref_st.finalize().solve(ref_st.cli_init_alphas(), 50, 0.0)
    for p in pends:
        out = p.result()
        assert out["info"] == info
        assert out["exit"] == ref["exit"] and out["n_iters"] == ref["n_iters"] and out["n_evals"] == ref["n_evals"]
        assert np.array_equal(out["alpha"], ref["alpha"])
        assert out["final_loss"] == ref["final_loss"]
        assert np.array_equal(out["trace"], ref["trace"])
    # a bad start (init_acc of a matrix whose counts sit in one column is fine; force many evaluations
    # instead through a tiny first batch): continue after the first wait
    st = device.DeviceStats(fx["K"])
    st.add_dense(X)
    p = device.PendingEstimate.__new__(device.PendingEstimate)
    uv = st.finalize()
    device.PendingEstimate.__init__(p, st, uv, 50, 0.0)
    assert p.launched >= 2
    out = p.result()
    assert np.array_equal(out["alpha"], ref["alpha"])


def test_enqueued_estimate_reports_data_errors():
    from bayespy_amd import device
    X = torch.tensor([[1, 2, 3], [4, 0, 600], [2, 2, 2]], dtype=torch.int32, device="cuda")
    st = device.DeviceStats(3, 10, 20)
    # bounds stated too small: counted as data errors by the pass, raised when the result is taken
    rc_st = st.add_dense(X, max_count=10, max_total=20)
    assert rc_st is st
    with pytest.raises(ValueError, match="statistics incomplete"):
        st.estimate_async(50, 0.0).result()
    # an all-zero column: the reference's ZeroDivisionError
    Z = torch.tensor([[1, 0, 3], [4, 0, 6], [2, 0, 2]], dtype=torch.int32, device="cuda")
    st = device.DeviceStats(3)
    st.add_dense(Z)
    with pytest.raises(ZeroDivisionError):
        st.estimate_async(50, 0.0).result()


@pytest.mark.parametrize("top", [16383, 16384, 16385, 16700, 40000])
def test_row_totals_around_the_wave_sum_limit(oracle, top):
    """The V partial sums are added inside every wave while they fit one (64 slots of 256 totals:
    row totals up to 16 384) and by the block reduction beyond: the same results either side of the
    limit, on all three launch structures, against the oracle."""
    from bayespy_amd import _lib, device
    rng = np.random.default_rng(top)
    K, N = 6, 3000
    p = rng.dirichlet(np.full(K, 0.7), size=N)
    totals = rng.integers(1, top + 1, size=N)
    totals[0] = top
    X = np.stack([rng.multinomial(int(t), q) for t, q in zip(totals, p)]).astype(np.int32)
    st = device.DeviceStats(K)
    st.add_dense(torch.from_numpy(X).cuda())
    assert st.check()["max_total"] == top
    uv = st.finalize()
    ref_st = oracle.append_rows(X)
    ref = oracle.find_priors(ref_st, ref_st.cli_init_alphas(), 50, 0.0)
    L = _lib.lib()
    outs = []
    try:
        for path in (2, 1, 3):  # single launch, launch pairs, one launch per evaluation
            _lib.check(L.dme_solve_set_path(path))
            outs.append(uv.solve(st.cli_init_alphas(), 50, 0.0))
    finally:
        _lib.check(L.dme_solve_set_path(0))
    for out in outs:
        np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=RTOL_ALPHA)
        assert out["n_iters"] == ref["n_iters"] and out["n_evals"] == ref["n_evals"]
        np.testing.assert_allclose(out["final_loss"], oracle.total_loss(ref["alpha"], ref_st, 0.0),
                                   rtol=RTOL_LOSS)
        assert np.array_equal(out["alpha"], outs[0]["alpha"]) and np.array_equal(out["trace"], outs[0]["trace"])
