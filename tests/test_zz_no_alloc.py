"""dme_eval_noalloc (the entry point that allocates nothing) against dme_eval, through the C ABI.

Both run the same kernels in the same order; only the home of the solver workspace differs (a
static device array inside the library against a caller-provided buffer), so the outputs must be
equal BIT FOR BIT.  Two calls in a row, the second on a smaller problem: the array keeps the first
call's contents and nothing may depend on them.  The oracle is the checker for the values
themselves (loss 1e-12, the bar of tests/test_gpu_solver.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _problem(rng, N, K, max_total):
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    X = np.stack([rng.multinomial(int(rng.integers(1, max_total)), rng.dirichlet(alpha))
                  for _ in range(N)]).astype(np.int32)
    X[:K] += np.eye(K, dtype=np.int32)  # no empty column
    return X


def _eval_both(uv, alpha, lam):
    """(scalars, grad, hdiag) from dme_eval and from dme_eval_noalloc, raw C-ABI calls."""
    from bayespy_amd import _lib
    L = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.from_numpy(np.asarray(alpha, dtype=np.float64)).cuda()
    outs = []
    for noalloc in (False, True):
        # poisoned outputs: a call that wrote nothing would be seen
        scal = torch.full((4,), float("nan"), dtype=torch.float64, device="cuda")
        grad = torch.full((uv.K,), float("nan"), dtype=torch.float64, device="cuda")
        hdiag = torch.full((uv.K,), float("nan"), dtype=torch.float64, device="cuda")
        common = (uv.Uf.data_ptr(), uv.ldU, uv.lenU.data_ptr(), uv.Vf.data_ptr(), uv.capV,
                  uv.lenV.data_ptr(), uv.K, a.data_ptr(), float(lam), scal.data_ptr(),
                  grad.data_ptr(), hdiag.data_ptr())
        if noalloc:
            _lib.check(L.dme_eval_noalloc(*common, stream))
        else:
            ws, wsb = uv.workspace()
            _lib.check(L.dme_eval(*common, ws, wsb, stream))
        torch.cuda.synchronize()
        outs.append((scal.cpu().numpy(), grad.cpu().numpy(), hdiag.cpu().numpy()))
    return outs


def test_noalloc_eval_equals_eval_on_two_calls_in_a_row(oracle):
    from bayespy_amd import device as dev
    rng = np.random.default_rng(77)
    # the second problem is smaller in every extent (K, ldU, lenV): it reuses the head of the array
    # the first call left behind
    for (N, K, max_total, lam) in ((4000, 300, 400, 0.0), (500, 7, 40, 0.25)):
        X = _problem(rng, N, K, max_total)
        st = dev.DeviceStats(K)
        st.add_dense(torch.from_numpy(X).cuda())
        st.check()
        uv = st.finalize()
        alpha = 0.1 + rng.random(K)
        (s0, g0, h0), (s1, g1, h1) = _eval_both(uv, alpha, lam)
        assert not np.isnan(s1).any() and not np.isnan(g1).any() and not np.isnan(h1).any()
        # same kernels, same order of operations: bit-identical
        assert s0.tobytes() == s1.tobytes()
        assert g0.tobytes() == g1.tobytes()
        assert h0.tobytes() == h1.tobytes()
        # and right: the oracle's dirichLogProb / gradient / Hessian at the same point
        ref = oracle.append_rows(X)
        np.testing.assert_allclose(s1[0], oracle.logprob(alpha, ref, lam), rtol=1e-12)
        np.testing.assert_allclose(s1[1], oracle.hconst(alpha, ref), rtol=1e-12)
        np.testing.assert_allclose(h1, oracle.hdiag(alpha, ref), rtol=1e-12)
        gref = np.asarray(oracle.gradient(alpha, ref, lam))
        np.testing.assert_allclose(g1, gref, rtol=1e-9, atol=1e-9 * (np.max(np.abs(gref)) + 1.0))

def test_noalloc_eval_rejects_shapes_beyond_its_scratch():
    from bayespy_amd import _lib
    L = _lib.lib()
    K, ldU, capV = 200_000, 600, 600
    assert L.dme_solver_workspace_bytes(K, ldU, capV) > (8 << 20)
    one = torch.zeros(8, dtype=torch.float64, device="cuda")
    p = one.data_ptr()
    # shape check comes before any pointer is touched
    rc = L.dme_eval_noalloc(p, ldU, p, p, capV, p, K, p, 0.0, p, 0, 0,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.ERR_WORKSPACE
    assert b"static scratch" in L.dme_last_error()
