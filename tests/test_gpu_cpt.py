"""GPU parity of the (Beta, W) conjugate hyperprior (dme_eval_cpt, dme_solve_cpt) against golden
vectors produced by the reference's ConjugatePriorTools module (tests/golden/make_golden_cpt.py),
and against oracle/cpt.py at a size the goldens do not cover.

Tolerances: alphas 1e-9 relative, loss 1e-12 relative.  digamma / trigamma are evaluated on the
device by recurrence + asymptotic series; the reference uses scipy's."""
import io

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

EXITS = {"A": 1, "B": 2, "C": 3}


def _cases():
    return list(range(len(load_golden("cpt_hyperprior")["cases"])))


def _uv(ds):
    from bayespy_amd import device
    return device.DeviceUV.from_lists(ds["U"], ds["V"])


@pytest.mark.parametrize("idx", _cases())
def test_cpt_functions(idx):
    fx = load_golden("cpt_hyperprior")
    c = fx["cases"][idx]
    uv = _uv(fx["datasets"][c["data"]])
    for f in c["funcs"]:
        lp, hc, grad, hd, _ = uv.evaluate(f["alpha"], 0.0, Beta=c["Beta"], W=c["W"])
        np.testing.assert_allclose(lp, f["logprob"], rtol=1e-12)
        np.testing.assert_allclose(hc, f["hconst"], rtol=1e-12)
        np.testing.assert_allclose(hd, f["hdiag"], rtol=1e-12)
        tsub = np.max(np.abs(f["grad"])) + np.max(np.abs(f["hdiag"])) ** 0.5 + 1.0
        np.testing.assert_allclose(grad, f["grad"], rtol=1e-9, atol=1e-12 * tsub)


@pytest.mark.parametrize("idx", _cases())
def test_cpt_solve(idx):
    fx = load_golden("cpt_hyperprior")
    c = fx["cases"][idx]
    s = c["solve"]
    uv = _uv(fx["datasets"][c["data"]])
    out = uv.solve(c["init"], 50, 0.0, Beta=c["Beta"], W=c["W"])
    assert out["exit"] == EXITS[s["exit"]]
    assert out["n_iters"] == len(s["trace"])
    np.testing.assert_allclose(out["alpha"], s["final"], rtol=1e-9)
    np.testing.assert_allclose(out["final_loss"], s["final_loss"], rtol=1e-12)
    np.testing.assert_allclose(out["trace"][:, 0], [t["loss"] for t in s["trace"]], rtol=1e-12)
    np.testing.assert_allclose(out["trace_alpha"], [t["priors"] for t in s["trace"]], rtol=1e-9)
    # An exit-B iteration walks the x0.9 loop until the trial loss stops exceeding the current
    # one, i.e. until their difference drops below the rounding noise of the loss sum.  Where that
    # happens depends on the summation order (sequential in the reference, compensated and
    # parallel here: 254 evaluations against 222 on the first case), so the count is compared
    # only for the runs that end by the gradient test.
    if s["exit"] == "A":
        assert out["n_evals"] == s["n_evals"]
    else:
        assert abs(out["n_evals"] - s["n_evals"]) <= 40 * len(s["trace"])


def test_cpt_module_and_cli():
    """The drop-in module and CLI with the reference's signatures."""
    from bayespy_amd.ConjugatePriorTools import dirichletMultinomialEstimation as CDME
    from bayespy_amd.ConjugatePriorTools import findDirichletPrior as cli
    fx = load_golden("cpt_hyperprior")
    ds = fx["datasets"]["zero_col_k8"]
    c = [x for x in fx["cases"] if x["data"] == "zero_col_k8"][1]
    data = CDME.CompressedRowData(8)
    for r in ds["rows"]:
        data.appendRow(r, 1)
    assert data.U == ds["U"] and data.V == ds["V"]
    f = c["funcs"][0]
    assert CDME.dirichLogProb(f["alpha"], data, c["Beta"], c["W"]) == pytest.approx(f["logprob"], rel=1e-12)
    assert CDME.getTotalLoss(f["alpha"], data, c["Beta"], c["W"]) == pytest.approx(f["loss"], rel=1e-12)
    np.testing.assert_allclose(CDME.priorHessianDiag(f["alpha"], data, c["W"]), f["hdiag"], rtol=1e-12)
    assert CDME.priorHessianConst(f["alpha"], data, c["W"]) == pytest.approx(f["hconst"], rel=1e-12)
    np.testing.assert_allclose(CDME.predictStepUsingHessian(f["grad"], f["alpha"], data, c["W"]),
                               f["step"], rtol=1e-10)
    np.testing.assert_allclose(CDME.predictStepLogSpace(f["grad"], f["alpha"], data, c["W"]),
                               f["stepLog"], rtol=1e-10)
    out = CDME.findDirichletPriors(data, list(c["init"]), 50, c["Beta"], c["W"])
    np.testing.assert_allclose(out, c["solve"]["final"], rtol=1e-9)

    text = "\n".join("\t".join(map(str, r)) for r in ds["rows"]) + "\n"
    H = ",".join(str(b) for b in c["Beta"]) + "," + str(c["W"])
    got = cli.main(["-K", "8", "-H", H, "-L", "ERROR"], stdin=io.StringIO(text), stdout=io.StringIO())
    np.testing.assert_allclose(got, c["solve"]["final"], rtol=1e-9)
    # without -H an all-zero column is refused (CPT/fdp.py:109-112)
    with pytest.raises(Exception, match="all 0s"):
        cli.main(["-K", "8", "-L", "ERROR"], stdin=io.StringIO(text), stdout=io.StringIO())


def test_cpt_k1000_vs_oracle():
    """K = 1000 with an informative hyperprior, against the numpy/scipy oracle."""
    from oracle import cpt
    from bayespy_amd import device
    fx = load_golden("dirmult_k1000")
    X = np.ascontiguousarray(fx["X"].astype(np.int32))
    st = device.DeviceStats(fx["K"])
    st.add_dense(torch.from_numpy(X).cuda())
    st.check()
    uv = st.finalize()
    U, V = uv.to_lists()
    d = cpt.UV(U, V)
    rng = np.random.default_rng(5)
    Beta = rng.random(fx["K"]) * 0.5
    W = 3.0
    a0 = st.cli_init_alphas().cpu().numpy()
    lp, hc, grad, hd, _ = uv.evaluate(a0, 0.0, Beta=Beta, W=W)
    np.testing.assert_allclose(lp, cpt.log_prob(a0, d, Beta, W), rtol=1e-12)
    np.testing.assert_allclose(hc, cpt.hconst(a0, d, W), rtol=1e-12)
    np.testing.assert_allclose(hd, cpt.hdiag(a0, d, W), rtol=1e-12)
    gref = cpt.gradient(a0, d, Beta, W)
    np.testing.assert_allclose(grad, gref, rtol=1e-9, atol=1e-9 * np.max(np.abs(gref)))
    ref = cpt.find_priors(d, a0, 50, Beta, W)
    out = uv.solve(a0, 50, 0.0, Beta=Beta, W=W)
    assert out["exit"] == EXITS[ref["exit"]] and out["n_iters"] == ref["n_iters"]
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
