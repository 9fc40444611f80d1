"""Pins the CPU oracle (oracle/dme_oracle.c) against golden vectors produced by the reference
itself (tests/golden/make_golden.py, DE variant = DirichletEstimation/).  Everything here is
bit-exact: the oracle keeps the reference's summation order and libm calls.  The NumPy-port
goldens ("NP") differ from DE only by summation order and are checked to 1e-11."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_append_rows_bit_exact(oracle, name):
    fx = load_golden(name)
    st = oracle.append_rows(fx["X"])
    assert [list(map(int, u)) for u in st.U] == fx["U"]
    assert list(map(int, st.V)) == fx["V"]
    # DE/fdp.py:87-89 accumulator and :104-108 normalisation, same order -> same bits
    assert st.init_acc.tolist() == fx["init_acc"]
    assert st.cli_init_alphas().tolist() == fx["alpha0_cli"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_closed_form_matches_loops(oracle, name):
    """U[k][n] = #{i: x_ik > n}, V[n] = #{i: sum_k x_ik > n}."""
    fx = load_golden(name)
    X = fx["X"].astype(np.int64)
    for k in range(fx["K"]):
        col = X[:, k]
        m = int(col.max())
        assert [int((col > n).sum()) for n in range(m)] == fx["U"][k]
    tot = X.sum(axis=1)
    assert [int((tot > n).sum()) for n in range(int(tot.max()))] == fx["V"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_function_level_bit_exact(oracle, name):
    fx = load_golden(name)
    st = oracle.stats_from_lists(fx["U"], fx["V"])
    for rec in fx["funcs"]:
        g = rec["DE"]
        a, hp = g["alpha"], g["hyperprior"]
        assert oracle.logprob(a, st, hp) == g["logprob"]
        assert oracle.total_loss(a, st, hp) == g["loss"]
        grad = oracle.gradient(a, st, hp)
        assert grad.tolist() == g["grad"]
        hc = oracle.hconst(a, st)
        hd = oracle.hdiag(a, st)
        assert hc == g["hconst"]
        assert hd.tolist() == g["hdiag"]
        assert oracle.sq_vector_size(grad) == g["gradSq"]
        if g["step"] is None:
            with pytest.raises(ZeroDivisionError):
                oracle.predicted_step(hc, hd, grad)
        else:
            assert oracle.predicted_step(hc, hd, grad).tolist() == g["step"]
            assert oracle.predicted_step_alt(hc, hd, grad, a).tolist() == g["stepAlt"]
        # NP variant: same maths, pairwise summation
        n = rec["NP"]
        np.testing.assert_allclose(oracle.total_loss(a, st, hp), n["loss"], rtol=1e-12)
        np.testing.assert_allclose(grad, n["grad"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_find_priors_bit_exact(oracle, name):
    fx = load_golden(name)
    st = oracle.stats_from_lists(fx["U"], fx["V"])
    n_checked = 0
    for s in fx["solves"]:
        if s["variant"] != "DE":
            continue
        if "raises" in s:
            with pytest.raises(ZeroDivisionError):
                oracle.find_priors(st, s["init"], s["iterations"], s["hyperprior"])
            n_checked += 1
            continue
        out = oracle.find_priors(st, s["init"], s["iterations"], s["hyperprior"])
        assert out["alpha"].tolist() == s["final"], (name, s["init_name"], s["hyperprior"])
        assert out["exit"] == s["exit"]
        assert out["n_iters"] == s["n_iters"]
        assert out["n_evals"] == s["n_evals"]
        assert out["trace_loss"].tolist() == [t["loss"] for t in s["trace"]]
        assert out["trace_gradsq"].tolist() == [t["gradSq"] for t in s["trace"]]
        assert out["trace_alpha"].tolist() == [t["priors"] for t in s["trace"]]
        if "evals" in s:
            m = min(len(out["eval_loss"]), len(s["evals"]))
            assert out["eval_loss"][:m].tolist() == [e["loss"] for e in s["evals"][:m]]
        assert oracle.total_loss(out["alpha"], st, s["hyperprior"]) == s["final_loss"]
        n_checked += 1
    assert n_checked > 0


def test_np_variant_close(oracle):
    """The NumPy port's answers (pairwise sums) stay within 1e-9 of the oracle's where the
    branch history is the same (exit + iteration count equal)."""
    for name in GOLDEN_CASES:
        fx = load_golden(name)
        st = oracle.stats_from_lists(fx["U"], fx["V"])
        de = {(s["init_name"], s["hyperprior"]): s for s in fx["solves"] if s["variant"] == "DE"}
        for s in fx["solves"]:
            if s["variant"] != "NP" or "raises" in s:
                continue
            d = de[(s["init_name"], s["hyperprior"])]
            if "raises" in d or d["n_evals"] != s["n_evals"] or d["exit"] == "C":
                continue
            np.testing.assert_allclose(d["final"], s["final"], rtol=1e-9)


def test_cpt_variant_agrees():
    """ConjugatePriorTools' original with no (Beta, W) prior gives the DE bits on test.csv."""
    fx = load_golden("test_csv")
    cpt = load_golden("cpt_test_csv")
    de = [s for s in fx["solves"] if s["variant"] == "DE" and s["init_name"] == "cli"
          and s["hyperprior"] == 0][0]
    assert cpt["final"] == de["final"]
    assert cpt["loss"] == de["final_loss"]


def test_csr_rows_match_dense(oracle):
    fx = load_golden("dirmult_k50")
    X = fx["X"].astype(np.int32)
    nz = X != 0
    rowptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))]).astype(np.int64)
    colidx = np.nonzero(nz)[1].astype(np.int32)
    vals = X[nz].astype(np.int32)
    st = oracle.append_rows_csr(rowptr, colidx, vals, fx["K"])
    assert [list(map(int, u)) for u in st.U] == fx["U"]
    assert list(map(int, st.V)) == fx["V"]
    assert st.init_acc.tolist() == fx["init_acc"]


def test_weighted_rows_bit_exact(oracle):
    """appendRow(row, w) with fractional, zero and unit weights mixed (reference-generated golden)."""
    import json, os
    from conftest import GOLDEN
    fx = json.load(open(os.path.join(GOLDEN, "weighted_rows.json")))
    X = np.asarray(fx["rows"], dtype=np.int32)
    w = np.asarray(fx["weights"], dtype=np.float64).copy()
    lo, hi = fx["unit_weight_span"]
    w[lo:hi] = 1.0
    st = oracle.append_rows_weighted(X, w)
    assert [u.tolist() for u in st.U] == fx["U"]
    assert st.V.tolist() == fx["V"]
    out = oracle.find_priors(st, fx["init"], 50, 0.0)
    assert out["alpha"].tolist() == fx["final"]
    assert out["n_iters"] == fx["n_iters"] and out["n_evals"] == fx["n_evals"]
    assert oracle.total_loss(out["alpha"], st, 0.0) == fx["final_loss"]


# ---- (Beta, W) conjugate hyperprior: oracle/cpt.py against the reference's CPT module ------------
def _cpt_cases():
    fx = load_golden("cpt_hyperprior")
    return fx, list(range(len(fx["cases"])))


@pytest.mark.parametrize("idx", _cpt_cases()[1])
def test_cpt_oracle_matches_reference(idx):
    from oracle import cpt
    fx = load_golden("cpt_hyperprior")
    c = fx["cases"][idx]
    ds = fx["datasets"][c["data"]]
    d = cpt.UV(ds["U"], ds["V"])
    for f in c["funcs"]:
        np.testing.assert_allclose(cpt.log_prob(f["alpha"], d, c["Beta"], c["W"]), f["logprob"], rtol=1e-13)
        np.testing.assert_allclose(cpt.total_loss(f["alpha"], d, c["Beta"], c["W"]), f["loss"], rtol=1e-13)
        np.testing.assert_allclose(cpt.gradient(f["alpha"], d, c["Beta"], c["W"]), f["grad"],
                                   rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(cpt.hconst(f["alpha"], d, c["W"]), f["hconst"], rtol=1e-13)
        np.testing.assert_allclose(cpt.hdiag(f["alpha"], d, c["W"]), f["hdiag"], rtol=1e-13)
    s = c["solve"]
    out = cpt.find_priors(d, c["init"], 50, c["Beta"], c["W"])
    assert out["exit"] == s["exit"] and out["n_iters"] == len(s["trace"])
    np.testing.assert_allclose(out["alpha"], s["final"], rtol=1e-9)
    np.testing.assert_allclose([t["loss"] for t in out["trace"]], [t["loss"] for t in s["trace"]],
                               rtol=1e-12)


def test_cpython_port_matches_the_c_oracle(oracle):
    """oracle/py_port.py (the timing stand-in for the reference's interpreted loops) builds the same
    U/V and mean accumulator as the C oracle, bit for bit."""
    from oracle import py_port
    fx = load_golden("dirmult_k50")
    X = fx["X"].astype(np.int32)[:400]
    data, acc = py_port.build(X.tolist(), fx["K"])
    ref = oracle.append_rows(X)
    assert data.U == [list(map(int, u)) for u in ref.U]
    assert data.V == list(map(int, ref.V))
    assert acc == ref.init_acc.tolist()


def test_numpy_structured_port_matches_the_c_oracle(oracle):
    """oracle/np_port.py (the timing stand-in for the reference's NumPy port: float64 ndarrays grown
    by np.append, a slice add per column per row) holds the same U/V values as the C oracle."""
    from oracle import np_port
    fx = load_golden("dirmult_k50")
    X = fx["X"].astype(np.int32)[:400]
    data, acc = np_port.build(X.tolist(), fx["K"])
    ref = oracle.append_rows(X)
    assert [u.tolist() for u in data.U] == [list(map(float, u)) for u in ref.U]
    assert data.V.tolist() == list(map(float, ref.V))
    assert acc == ref.init_acc.tolist()
