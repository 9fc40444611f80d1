"""Ragged statistics on the device: the overflow store of the statistics passes,
the compact form, the ragged suffix scan and the solver on the ragged layout.

The reference's U[k] is as long as column k's own maximum and V as long as the largest row total
(DE/dme.py:118, :122); here the dense-padded and the ragged layouts must give the same lists bit for
bit, and the solver - same device functions, same order of operations - the same bits on both."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    from bayespy_amd import device
    return device


def _lists_equal(uv, st_oracle):
    U, V = uv.to_lists()
    assert U == [list(map(int, u)) for u in st_oracle.U]
    assert V == list(map(int, st_oracle.V))


def _same_solution(a, b):
    assert np.array_equal(a["alpha"], b["alpha"])
    assert a["final_loss"] == b["final_loss"]
    assert (a["exit"], a["n_iters"], a["n_evals"]) == (b["exit"], b["n_iters"], b["n_evals"])
    assert np.array_equal(a["trace"], b["trace"])


@pytest.mark.parametrize("name", ["test_csv", "one_doublesided", "heavy_k20", "dirmult_k50",
                                  "dirmult_k1000"])
def test_ragged_layout_equals_dense_layout(name, oracle):
    """Scan + solver on both layouts: identical lists, identical evaluation, identical Newton trace -
    with the scalar hyperprior, with the (Beta, W) one, from the CLI start and from [1.0] * K (the
    fallback branches)."""
    dev = _dev()
    fx = load_golden(name)
    X = torch.from_numpy(np.ascontiguousarray(fx["X"].astype(np.int32))).cuda()
    K = fx["K"]
    st = dev.DeviceStats(K)
    st.add_dense(X)
    st.check()
    dense, rag = st.finalize(ragged=False), st.finalize(ragged=True)
    assert rag.ragged is not None and dense.ragged is None
    assert rag.to_lists() == dense.to_lists() == (fx["U"], fx["V"])
    assert torch.equal(rag.dense_U()[:, :dense.ldU], dense.dense_U()[:, :rag.dense_U().shape[1]])
    # memory: the ragged U holds sum(len(U[k])) terms
    assert rag.Uf.numel() == max(sum(len(u) for u in fx["U"]), 1)
    a0 = st.cli_init_alphas().cpu().numpy()
    for lam in (0.0, 0.01):
        ed, er = dense.evaluate(a0, lam), rag.evaluate(a0, lam)
        assert ed[0] == er[0] and ed[1] == er[1] and ed[4] == er[4]
        assert np.array_equal(ed[2], er[2]) and np.array_equal(ed[3], er[3])
    _same_solution(dense.solve(a0, 50, 0.0), rag.solve(a0, 50, 0.0))
    _same_solution(dense.solve([1.0] * K, 50, 0.0), rag.solve([1.0] * K, 50, 0.0))
    beta = np.linspace(0.1, 0.3, K)
    ed, er = dense.evaluate(a0, Beta=beta, W=0.5), rag.evaluate(a0, Beta=beta, W=0.5)
    assert ed[0] == er[0] and np.array_equal(ed[2], er[2]) and np.array_equal(ed[3], er[3])
    _same_solution(dense.solve(a0, 30, Beta=beta, W=0.5), rag.solve(a0, 30, Beta=beta, W=0.5))
    # the pipelined entry (dme_solve_enqueue_ragged) against the waiting one
    o = st.estimate_async(50).result()
    ref = dense.solve(a0, 50, 0.0)
    assert np.array_equal(o["alpha"], ref["alpha"]) and o["n_evals"] == ref["n_evals"]


@pytest.mark.parametrize("dtype", ["u16", "i32"])
@pytest.mark.parametrize("K,N,dense_bins", [(2, 50_000, 16), (4, 20_000, 8), (20, 6_000, 16),
                                            (100, 4_000, 32), (1000, 3_000, 4), (3000, 700, 6)])
def test_overflow_store_of_the_dense_passes(K, N, dense_bins, dtype, oracle):
    """Counts beyond the dense bins become records: the lists are the oracle's, whichever dense kernel
    the plan picks (pair counts, side-by-side rows, wave per row, teams) and whichever of its rare
    paths a count takes (beyond the LDS bins but dense; beyond the dense bins; >= 65536 for int32)."""
    dev = _dev()
    rng = np.random.default_rng(K * 7 + dense_bins)
    X = (rng.integers(1, 12, size=(N, K)) * (rng.random((N, K)) < min(0.5, 8.0 / K))).astype(np.int64)
    hot = rng.integers(0, N, size=40)
    X[hot, rng.integers(0, K, size=40)] = rng.integers(20, 5000, size=40)
    if dtype == "i32" and K <= 100:  # (the oracle holds U dense: K x the largest count)
        X[hot[0], 0] = 70_000
        X[hot[1], K - 1] = 300_000
    npdt = np.uint16 if dtype == "u16" else np.int32
    Xd = torch.from_numpy(np.ascontiguousarray(X.astype(npdt))).cuda()
    st = dev.DeviceStats(K, dense_bins=dense_bins)
    st.add_dense(Xd)
    info = st.check()
    assert st.overflow is not None and st.nbins == dense_bins
    assert info["rows"] == N and info["max_count"] == int(X.max())
    ref = oracle.append_rows(X.astype(npdt))
    uv = st.finalize()
    _lists_equal(uv, ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)
    # two halves, merged (dme_overflow_append), and a save / load round trip
    a, b = dev.DeviceStats(K, dense_bins=dense_bins), dev.DeviceStats(K, dense_bins=dense_bins)
    a.add_dense(Xd[:N // 2])
    b.add_dense(Xd[N // 2:])
    a.merge_(b)
    _lists_equal(a.finalize(), ref)


@pytest.mark.parametrize("dtype,K", [("u16", 3000), ("u16", 9000), ("u16", 20_000), ("i32", 1000),
                                     ("i32", 3000), ("i32", 6000), ("i32", 12_000)])
def test_overflow_records_of_wide_rows(dtype, K):
    """Rows that teams of waves share (and int32 rows of the packed kernels): cells planted beyond the
    dense bins - below and, for int32, above 65535, several in one 8-column chunk of one row - come back
    as exactly those records; the dense bins hold the rest.  No oracle: its dense U would be K x the
    largest count."""
    dev = _dev()
    N, dense_bins = 1500, 8
    rng = np.random.default_rng(K)
    X = np.zeros((N, K), dtype=np.int64)
    sel = rng.random((N, K)) < 6.0 / K
    X[sel] = rng.integers(1, 12, size=int(sel.sum()))
    big_hi = 60_000 if dtype == "u16" else 400_000
    rows = rng.integers(0, N, size=60)
    cols = rng.integers(0, K, size=60)
    X[rows, cols] = rng.integers(dense_bins, big_hi, size=60)
    base = (int(cols[0]) // 8) * 8  # three in a row's one chunk
    X[rows[0], base:base + 3] = [big_hi - 1, dense_bins, big_hi - 2]
    npdt = np.uint16 if dtype == "u16" else np.int32
    Xd = torch.from_numpy(np.ascontiguousarray(X.astype(npdt))).cuda()
    st = dev.DeviceStats(K, dense_bins=dense_bins, overflow_capacity=4096)
    st.add_dense(Xd)
    info = st.check()
    assert info["rows"] == N and info["max_count"] == int(X.max())
    ok, ov, on = st._overflow_records()
    got = {}
    for k, v, n in zip(ok.tolist(), ov.tolist(), on.tolist()):
        got[(k, v)] = got.get((k, v), 0) + n
    rr, cc = np.nonzero(X >= dense_bins)
    want = {}
    for k, v in zip(cc.tolist(), X[rr, cc].tolist()):
        want[(k, v)] = want.get((k, v), 0) + 1
    assert got == want
    hist = st.hist.view(K, st.nbins).cpu().numpy()
    for v in range(1, dense_bins):
        assert np.array_equal(hist[:, v], (X == v).sum(axis=0)), v
    w = 1.0 / (1.0 + X.sum(axis=1))
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), (X * w[:, None]).sum(axis=0), rtol=1e-12, atol=1e-300)


def test_overflow_store_save_load_and_full_store(tmp_path, oracle):
    dev = _dev()
    rng = np.random.default_rng(5)
    K, N = 30, 5000
    X = (rng.integers(1, 400, size=(N, K)) * (rng.random((N, K)) < 0.2)).astype(np.int32)
    Xd = torch.from_numpy(X).cuda()
    st = dev.DeviceStats(K, dense_bins=10)
    st.add_dense(Xd)
    ref = oracle.append_rows(X)
    path = str(tmp_path / "stats.npz")
    st.save(path)
    back = dev.DeviceStats.load(path)
    assert back.overflow is not None
    _lists_equal(back.finalize(), ref)
    more = dev.DeviceStats(K, dense_bins=10)
    more.add_dense(Xd)
    back.merge_(more)
    ref2 = oracle.append_rows(np.concatenate([X, X]))
    _lists_equal(back.finalize(), ref2)
    # a store that is too small says so, and says how to fix it
    tiny = dev.DeviceStats(K, dense_bins=10, overflow_capacity=100)
    tiny.add_dense(Xd)
    with pytest.raises(ValueError, match="overflow_capacity"):
        tiny.check()


@pytest.mark.parametrize("K,partitioned", [(700, True), (50_000, True), (600_000, True), (300, False)])
def test_overflow_store_of_the_csr_passes(K, partitioned, oracle):
    """The CSR entry on all its paths (atomics; the tile / row-wise design; the first design beyond
    65 536 columns) with counts on both sides of the dense bins and of the 256 a record holds."""
    dev = _dev()
    rng = np.random.default_rng(K)
    N = 30_000 if partitioned else 400
    rows = [np.unique(rng.integers(0, K, size=n)) for n in rng.integers(1, 60, size=N)]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(r) for r in rows])
    nnz = int(rowptr[N])
    col = np.concatenate(rows).astype(np.int32)
    val = rng.integers(1, 9, size=nnz).astype(np.int32)
    big = rng.integers(0, nnz, size=300)
    val[big] = rng.integers(10, 3000, size=300)
    val[big[:3]] = [100_000, 70_000, 255]
    dense_bins = 12
    st = dev.DeviceStats(K, dense_bins=dense_bins)
    st.add_csr(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda())
    info = st.check()
    assert info["rows"] == N and info["max_count"] == 100_000 and st.nbins == dense_bins
    uv = st.finalize()
    # independent reference: value histogram per (column, count) and per row total with numpy
    tot = np.add.reduceat(val.astype(np.int64), rowptr[:-1])
    U, V = uv.to_lists()
    histV = np.bincount(tot)
    Vref = np.cumsum(histV[::-1])[::-1][1:]
    assert V == [int(x) for x in Vref]
    order = np.lexsort((val, col))
    cs, vs = col[order], val[order]
    start = np.searchsorted(cs, np.arange(K), "left")
    end = np.searchsorted(cs, np.arange(K), "right")
    for k in rng.choice(K, size=min(K, 400), replace=False).tolist() + col[big[:3]].tolist():
        v = vs[start[k]:end[k]].astype(np.int64)
        if v.size == 0:
            assert U[k] == []
            continue
        h = np.bincount(v)
        assert U[k] == [int(x) for x in np.cumsum(h[::-1])[::-1][1:]], k
    assert uv.Uf.numel() == max(sum(len(u) for u in U), 1)


def test_one_outlier_count_at_fifty_thousand_columns_stays_under_two_gigabytes():
    """VERDICT r2 item 3's bar: K = 50 000 with one count of 100 000 estimates in < 2 GB (dense
    K x (max_count + 1) statistics would ask for 120 GB), and the outlier changes exactly its own
    column and the row-total vector."""
    dev = _dev()
    rng = np.random.default_rng(11)
    K, N0 = 50_000, 40_000
    rows = [np.unique(rng.integers(0, K, size=n)) for n in rng.integers(20, 120, size=N0)]
    # every column appears at least once (the reference divides by zero on an empty column, DE/dme.py:69)
    rows += [np.arange(j, j + 50) for j in range(0, K, 50)]
    N = len(rows)
    rowptr = np.zeros(N + 2, dtype=np.int64)
    rowptr[1:N + 1] = np.cumsum([len(r) for r in rows])
    nnz = int(rowptr[N])
    col = np.concatenate(rows).astype(np.int32)
    val = rng.integers(1, 6, size=nnz).astype(np.int32)
    # the extra row: one entry, column 7, count 100 000
    rowptr[N + 1] = nnz + 1
    col2 = np.concatenate([col, np.array([7], dtype=np.int32)])
    val2 = np.concatenate([val, np.array([100_000], dtype=np.int32)])
    rp, ci, va = (torch.from_numpy(a).cuda() for a in (rowptr, col2, val2))
    plain = dev.DeviceStats(K)
    plain.add_csr(rp[:N + 1], ci[:nnz], va[:nnz])
    out0 = plain.estimate_async(50).result()
    U0, V0 = out0["uv"].to_lists()
    del plain, out0
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    st = dev.DeviceStats(K)
    st.add_csr(rp, ci, va)
    out = st.estimate_async(50).result()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 2 * 2**30, peak
    assert st.overflow is not None and out["info"]["max_count"] == 100_000
    assert out["uv"].ragged is not None
    U1, V1 = out["uv"].to_lists()
    assert len(U1[7]) == 100_000
    assert U1[7][:len(U0[7])] == [u + 1 for u in U0[7]] and set(U1[7][len(U0[7]):]) == {1}
    assert all(U1[k] == U0[k] for k in range(K) if k != 7)
    assert len(V1) == 100_000 and V1[:len(V0)] == [v + 1 for v in V0] and set(V1[len(V0):]) == {1}
    assert out["exit"] in (1, 2, 3) and np.all(np.isfinite(out["alpha"])) and np.all(out["alpha"] > 0)


def test_heavy_tail_config_runs_ragged(oracle):
    """BASELINE configs[4] shape (K = 100, row totals to 1e5) at 20 000 rows: the statistics are laid
    out ragged (sum of the columns' own lengths instead of 100 x 1e5 cells) and the solve equals the
    oracle's to the bars of the dense path (1e-9 on alpha, 1e-12 on the loss)."""
    dev = _dev()
    rng = np.random.default_rng(20260005)
    K, N = 100, 20_000
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    tot = np.minimum(np.floor(5 * (rng.pareto(1.2, size=N) + 1)), 100_000).astype(np.int64)
    tot[0] = 100_000
    p = rng.dirichlet(alpha, size=N)
    X = np.stack([rng.multinomial(t, pi) for t, pi in zip(tot, p)]).astype(np.int32)
    X[0] = 0
    X[0, 3] = 100_000  # the largest total in one cell: 100 x 100 001 dense cells
    st = dev.DeviceStats(K)
    st.add_dense(torch.from_numpy(X).cuda())
    st.check()
    uv = st.finalize()
    assert uv.ragged is not None and st.overflow is None
    ref = oracle.append_rows(X)
    _lists_equal(uv, ref)
    assert uv.Uf.numel() < K * 100_000 / 4
    a0 = st.cli_init_alphas().cpu().numpy()
    got = uv.solve(a0, 50, 0.0)
    want = oracle.find_priors(ref, a0, 50, 0.0)
    np.testing.assert_allclose(got["alpha"], want["alpha"], rtol=1e-9)
    want_loss = oracle.total_loss(want["alpha"], ref, 0.0)
    assert abs(got["final_loss"] - want_loss) <= 1e-12 * abs(want_loss)
    if got["fragile_decisions"] == 0:
        assert (got["n_iters"], got["n_evals"]) == (want["n_iters"], want["n_evals"])


def test_foreign_lists_of_very_different_lengths_stay_ragged():
    """DeviceUV.from_lists (data objects that carry their own .U / .V): one long list among many short
    ones is uploaded ragged and evaluates like the dense-padded upload."""
    dev = _dev()
    rng = np.random.default_rng(2)
    K = 64
    lens = rng.integers(1, 30, size=K)
    lens[5] = 80_000
    U = [sorted(rng.integers(1, 1000, size=n).tolist(), reverse=True) for n in lens]
    V = sorted(rng.integers(1, 5000, size=90_000).tolist(), reverse=True)
    rag = dev.DeviceUV.from_lists(U, V)
    assert rag.ragged is not None and rag.Uf.numel() == int(lens.sum())
    assert rag.to_lists() == ([list(map(float, u)) for u in U], list(map(float, V)))
    host = np.zeros((K, int(lens.max())))
    for k, u in enumerate(U):
        host[k, :len(u)] = u
    dense = dev.DeviceUV(K, torch.from_numpy(host).cuda().view(-1), int(lens.max()),
                         torch.from_numpy(lens.astype(np.int64)).cuda(),
                         torch.tensor(V, dtype=torch.float64, device="cuda"),
                         torch.tensor([len(V)], dtype=torch.int64, device="cuda"))
    a = 0.2 + rng.random(K)
    ed, er = dense.evaluate(a), rag.evaluate(a)
    assert ed[0] == er[0] and ed[1] == er[1]
    assert np.array_equal(ed[2], er[2]) and np.array_equal(ed[3], er[3])
