"""Weighted rows on the device (appendRow(row, w), DE/dme.py:113-123 with w != 1; the mixture EM of
CPT/dirichletMixtureEstimation.py:128-133 feeds C of them per row) against the oracle's literal
loops: U/V to 1e-12 relative (fp64 sums in another order), lengths exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _rows(rng, N, K, vmax, density):
    X = rng.integers(1, vmax + 1, size=(N, K)) * (rng.random((N, K)) < density)
    return X.astype(np.int32)


def _compare(uv, ref):
    U, V = uv.to_lists()
    assert [len(u) for u in U] == [len(u) for u in ref.U]
    assert len(V) == len(ref.V)
    for got, want in zip(U, ref.U):
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(V, ref.V, rtol=1e-12, atol=1e-300)


# K = 40: every count below the LDS bin limit; K = 1000 with counts to 90: most counts above it
# (global path) and row totals beyond the LDS total bins; K = 20000: column maxima kept in global
# memory, no LDS count bins at all
@pytest.mark.parametrize("N,K,vmax,density", [(3000, 40, 30, 0.6), (700, 1000, 90, 0.3),
                                              (64, 20000, 5, 0.01), (1, 7, 3, 1.0)])
def test_weighted_rows_vs_oracle(oracle, N, K, vmax, density):
    from bayespy_amd import device
    rng = np.random.default_rng(N + K)
    X = _rows(rng, N, K, vmax, density)
    w = rng.random(N) * 10.0 ** rng.integers(-3, 3, size=N)
    w[::7] = 0.0   # a zero weight still lengthens U[k] (DE/dme.py:118)
    ws = device.WeightedStats(K)
    ws.add_dense(torch.from_numpy(X).cuda(), torch.from_numpy(w).cuda(),
                 int(X.max()), int(X.sum(axis=1).max()))
    ws.check()
    _compare(ws.to_uv(), oracle.append_rows_weighted(X, w))


def test_weighted_batch_equals_one_pass_per_weight_column(oracle):
    """C weight columns in one pass = C separate passes (and the oracle's loops), chunked input."""
    from bayespy_amd import device
    rng = np.random.default_rng(5)
    N, K, C = 5000, 96, 5
    X = _rows(rng, N, K, 60, 0.4)
    R = rng.dirichlet(np.ones(C), size=N)  # responsibilities: rows sum to 1
    ws = device.WeightedStats(K, C=C)
    mc, mt = int(X.max()), int(X.sum(axis=1).max())
    Xd, Rd = torch.from_numpy(X).cuda(), torch.from_numpy(R).cuda()
    for lo in range(0, N, 1800):
        ws.add_dense(Xd[lo:lo + 1800], Rd[lo:lo + 1800], mc, mt)
    ws.check()
    assert int(ws.status[0]) == N
    for c in range(C):
        _compare(ws.to_uv(c), oracle.append_rows_weighted(X, R[:, c]))
    # the weights of a row sum to 1: the C statistics add up to the unweighted ones
    st = device.DeviceStats(K)
    st.add_dense(Xd)
    Uint, Vint = st.finalize().to_lists()
    total = ws.to_uv(0)
    for c in range(1, C):
        ws.add_into(total, c)
    Usum, Vsum = total.to_lists()
    for got, want in zip(Usum, Uint):
        np.testing.assert_allclose(got, want, rtol=1e-12)
    np.testing.assert_allclose(Vsum, Vint, rtol=1e-12)


def test_weighted_bounds_and_negatives_are_reported():
    from bayespy_amd import device
    X = torch.tensor([[1, 2, 3], [4, 0, 9]], dtype=torch.int32, device="cuda")
    w = torch.tensor([0.5, 0.25], dtype=torch.float64, device="cuda")
    ws = device.WeightedStats(3)
    ws.add_dense(X, w, 4, 13)  # the 9 does not fit
    with pytest.raises(ValueError):
        ws.check()
    ws = device.WeightedStats(3)
    ws.add_dense(X, w, 9, 10)  # the total 13 does not fit
    with pytest.raises(ValueError):
        ws.check()
    X[0, 1] = -2
    ws = device.WeightedStats(3)
    ws.add_dense(X, w, 9, 13)
    with pytest.raises(ValueError):
        ws.check()
    with pytest.raises(ValueError):
        device.WeightedStats(3, C=2).add_dense(X, w, 9, 13)  # one weight per row, two expected


@pytest.mark.parametrize("N,K,C,vmax,density", [(40_000, 100, 5, 60, 0.3), (20_000, 300, 2, 12, 0.2),
                                                (10_000, 1500, 1, 4, 0.05), (9000, 64, 8, 200, 0.5)])
def test_weighted_images_path_vs_oracle_and_atomics_path(oracle, N, K, C, vmax, density):
    """From 32 workgroups on (8192 rows) and with a workspace the workgroups leave their LDS bins as images
    and weighted_reduce_kernel adds them up (dme_stats_dense_weighted_batch_ws_i32); without a workspace they
    send them to the global histograms with atomics.  Both against the oracle's loops (DE/dme.py:113-123
    with weights), on top of statistics that are already there (two calls), and against each other."""
    from bayespy_amd import device, _lib
    rng = np.random.default_rng(N + K + C)
    X = _rows(rng, N, K, vmax, density)
    X[5, 3] = vmax + 40   # beyond what most rows hold: the global-atomics path beside the images
    R = rng.dirichlet(np.ones(C), size=N) if C > 1 else rng.random((N, 1))
    mc, mt = int(X.max()), int(X.sum(axis=1).max())
    Xd, Rd = torch.from_numpy(X).cuda(), torch.from_numpy(np.ascontiguousarray(R)).cuda()
    ws = device.WeightedStats(K, C=C)
    half = N // 2 + 37
    ws.add_dense(Xd[:half], Rd[:half], mc, mt)     # N // 2 + 37 >= 4500 rows ... images only from 8192 on
    ws.add_dense(Xd, Rd, mc, mt)                   # the whole matrix on top: every row counted twice more
    ws.check()
    assert ws._ws is not None and int(ws.status[0]) == N + half
    # the same through the raw entry without a workspace
    L = _lib.lib()
    ref = device.WeightedStats(K, C=C)
    ref.reserve(mc, mt)
    for lo, hi in ((0, half), (0, N)):
        _lib.check(L.dme_stats_dense_weighted_batch_i32(
            Xd[lo:hi].data_ptr(), Rd[lo:hi].data_ptr(), C, hi - lo, K, Xd.stride(0), ref.histW.data_ptr(),
            ref.nbins, ref.histVW.data_ptr(), ref.tbins, ref.maxcol.data_ptr(), ref.maxtot.data_ptr(),
            ref.status.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    np.testing.assert_allclose(ws.histW.cpu().numpy(), ref.histW.cpu().numpy(), rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(ws.histVW.cpu().numpy(), ref.histVW.cpu().numpy(), rtol=1e-12, atol=1e-300)
    assert torch.equal(ws.maxcol, ref.maxcol) and torch.equal(ws.maxtot, ref.maxtot)
    X2 = np.concatenate([X[:half], X])
    uvs = ws.to_uvs()   # all C components from one launch: the same bits as one launch per component
    for c in range(C):
        one = ws.to_uv(c)
        _compare(one, oracle.append_rows_weighted(X2, np.concatenate([R[:half, c], R[:, c]])))
        assert torch.equal(uvs[c].Uf, one.Uf) and torch.equal(uvs[c].Vf, one.Vf)
        assert torch.equal(uvs[c].lenU, one.lenU) and torch.equal(uvs[c].lenV, one.lenV)
    # bit-reproducible for the same images? no: LDS atomics inside a workgroup add in arrival order.


def test_weighted_workspace_entry_edge_cases(oracle):
    """dme_stats_dense_weighted_batch_ws_i32 through the raw C ABI: a workspace that is too small (the pass
    of the entry without one, same statistics), a misaligned one (refused), no rows (nothing happens), and
    the workspace query for shapes it does not take (0)."""
    from bayespy_amd import device, _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    N, K, C = 12_000, 50, 3
    X = _rows(rng, N, K, 9, 0.5)
    R = rng.dirichlet(np.ones(C), size=N)
    mc, mt = int(X.max()), int(X.sum(axis=1).max())
    Xd, Rd = torch.from_numpy(X).cuda(), torch.from_numpy(R).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def run(ws_ptr, ws_bytes, n=N):
        st = device.WeightedStats(K, C=C)
        st.reserve(mc, mt)
        rc = L.dme_stats_dense_weighted_batch_ws_i32(
            Xd.data_ptr(), Rd.data_ptr(), C, n, K, Xd.stride(0), st.histW.data_ptr(), st.nbins,
            st.histVW.data_ptr(), st.tbins, st.maxcol.data_ptr(), st.maxtot.data_ptr(), st.status.data_ptr(),
            ws_ptr, ws_bytes, stream)
        torch.cuda.synchronize()
        return rc, st

    need = int(L.dme_stats_weighted_workspace_bytes(K, C, mc + 1, mt + 1))
    assert need > 0 and L.dme_stats_weighted_workspace_bytes(K, 65, 4, 4) == 0
    ws = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
    rc_full, full = run(ws.data_ptr(), need)
    rc_small, small = run(ws.data_ptr(), 4096)          # too small: atomics, not an error
    rc_none, none = run(None, 0)
    assert rc_full == 0 and rc_small == 0 and rc_none == 0
    for other in (small, none):
        np.testing.assert_allclose(full.histW.cpu().numpy(), other.histW.cpu().numpy(), rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(full.histVW.cpu().numpy(), other.histVW.cpu().numpy(), rtol=1e-12, atol=1e-300)
    for c in range(C):
        _compare(full.to_uv(c), oracle.append_rows_weighted(X, R[:, c]))
    rc_bad, _ = run(ws.data_ptr() + 4, need)            # 4-byte aligned only
    assert rc_bad != 0 and b"aligned" in L.dme_last_error()
    rc_empty, empty = run(ws.data_ptr(), need, n=0)
    assert rc_empty == 0 and float(empty.histW.abs().sum()) == 0.0 and int(empty.status[0]) == 0


def test_weighted_images_path_at_full_grid_against_torch_reductions():
    """300 000 rows (one workgroup per CU, the bench's grid) of K = 100 with five weights per row: every
    (component, column, count) cell and every (component, row total) cell against fp64 index_add_
    reductions by torch on the device, to 1e-11 (sums of the same weights in another order)."""
    from bayespy_amd import device
    N, K, C, M = 300_000, 100, 5, 100
    rng = np.random.default_rng(3)
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    g = torch.Generator(device="cuda").manual_seed(9)
    totals = torch.randint(1, M + 1, (N,), dtype=torch.int32, device="cuda", generator=g)
    X = device.synth_dirmult(N, K, alpha, totals, seed=9, dtype=torch.int32, device=torch.device("cuda", 0))
    R = torch.rand((N, C), dtype=torch.float64, device="cuda", generator=g)
    R /= R.sum(dim=1, keepdim=True)
    ws = device.WeightedStats(K, C=C)
    ws.add_dense(X, R, M, M)
    ws.check()
    nb, tb = ws.nbins, ws.tbins
    Xl = X.long()
    cell = (torch.arange(K, device="cuda") * nb)[None, :] + Xl           # [N, K] index of (column, count)
    tot = Xl.sum(dim=1)
    got_h = ws.histW.view(C, K, nb)
    got_v = ws.histVW.view(C, tb)
    for c in range(C):
        ref = torch.zeros(K * nb, dtype=torch.float64, device="cuda")
        ref.index_add_(0, cell.reshape(-1), R[:, c:c + 1].expand(N, K).reshape(-1))
        ref = ref.view(K, nb)
        torch.testing.assert_close(got_h[c][:, 1:], ref[:, 1:], rtol=1e-11, atol=1e-13)
        refv = torch.zeros(tb, dtype=torch.float64, device="cuda")
        refv.index_add_(0, tot, R[:, c])
        torch.testing.assert_close(got_v[c][1:], refv[1:], rtol=1e-11, atol=1e-13)
    assert int(ws.status[0]) == N and int(ws.maxtot[0]) == int(tot.max())
    assert torch.equal(ws.maxcol, Xl.max(dim=0).values)
