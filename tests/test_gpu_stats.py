"""GPU parity of the statistics build (dme_stats_dense_*, dme_stats_csr_i32, dme_suffix_scan)
against the CPU oracle and the reference-generated golden fixtures.  Integer results are
bit-exact; the fp64 mean accumulator differs from the reference only by summation order
(tolerance 1e-12 relative, stated per assert)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    from bayespy_amd import device
    return device


def _to_dev(X, dtype):
    if dtype == "u16":
        return torch.from_numpy(np.ascontiguousarray(X.astype(np.uint16))).cuda()
    return torch.from_numpy(np.ascontiguousarray(X.astype(np.int32))).cuda()


def _build(X, dtype, **kw):
    dev = _dev()
    st = dev.DeviceStats(X.shape[1])
    st.add_dense(_to_dev(X, dtype), **kw)
    info = st.check()
    return st, st.finalize(), info


def _assert_uv_equal(uv, st_oracle):
    U, V = uv.to_lists()
    assert U == [list(map(int, u)) for u in st_oracle.U]
    assert V == list(map(int, st_oracle.V))


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("dtype", ["i32", "u16"])
def test_golden_stats_bit_exact(name, dtype):
    fx = load_golden(name)
    X = fx["X"]
    if dtype == "u16" and X.max() > 65535:
        pytest.skip("does not fit uint16")
    st, uv, info = _build(X, dtype)
    U, V = uv.to_lists()
    assert U == fx["U"]
    assert V == fx["V"]
    assert info["rows"] == fx["N"]
    assert info["max_count"] == int(X.max())
    assert info["max_total"] == int(X.astype(np.int64).sum(axis=1).max())
    # DE/fdp.py:87-89 accumulator: same products, different summation order -> 1e-12 relative
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), fx["init_acc"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(st.cli_init_alphas().cpu().numpy(), fx["alpha0_cli"], rtol=1e-12)


SHAPES = [
    # (N, K, max value, density)
    (1, 1, 5, 1.0), (7, 1, 3, 0.5), (64, 2, 60, 0.9), (65, 3, 9, 0.7), (129, 5, 300, 0.6),
    (1000, 7, 20, 0.3), (333, 8, 70000, 0.5), (257, 16, 12, 0.2), (500, 20, 100, 0.3),
    (100, 63, 4, 0.1), (100, 64, 4, 0.1), (100, 65, 4, 0.1), (300, 100, 50, 0.1),
    (200, 128, 1000, 0.05), (200, 129, 9, 0.05), (150, 250, 30, 0.1), (90, 512, 7, 0.1),
    (120, 1000, 400, 0.07), (60, 1024, 3, 0.1), (40, 2048, 90, 0.02), (30, 4096, 5, 0.01),
    (4500, 1000, 120, 0.05),
]


@pytest.mark.parametrize("N,K,vmax,density", SHAPES)
@pytest.mark.parametrize("dtype", ["i32", "u16"])
def test_random_shapes_vs_oracle(oracle, N, K, vmax, density, dtype):
    if dtype == "u16" and vmax > 65535:
        vmax = 65535
    rng = np.random.default_rng(N * 1000 + K)
    X = rng.integers(1, vmax + 1, size=(N, K)) * (rng.random((N, K)) < density)
    # a few large values to exercise the global-atomic path beside the LDS bins
    X[rng.integers(0, N), rng.integers(0, K)] = vmax
    ref = oracle.append_rows(X.astype(np.uint16) if dtype == "u16" else X.astype(np.int32))
    st, uv, info = _build(X, dtype)
    _assert_uv_equal(uv, ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)


def test_padded_rows_and_offsets(oracle):
    """ld > K and a base pointer that is only element-aligned (narrower vector loads)."""
    dev = _dev()
    rng = np.random.default_rng(5)
    for K, ld, off in [(1000, 1008, 0), (1000, 1003, 0), (100, 100, 1), (20, 24, 2), (50, 64, 3),
                       # int32: packed kernel (16-byte rows, K % 8 == 0) and its neighbours
                       (1000, 1000, 2), (1024, 1028, 0), (512, 516, 4), (4096, 4100, 0), (1000, 1002, 0)]:
        N = 300
        big = rng.integers(0, 40, size=(N, ld)) * (rng.random((N, ld)) < 0.2)
        for dtype, npdt in (("u16", np.uint16), ("i32", np.int32)):
            flat = torch.zeros(N * ld + 8, dtype=torch.int32 if dtype == "i32" else torch.uint16,
                               device="cuda")
            host = torch.from_numpy(big.astype(npdt).reshape(-1))
            flat[off:off + N * ld] = host.cuda()
            view = flat[off:off + N * ld].view(N, ld)[:, :K]
            st = dev.DeviceStats(K)
            st.add_dense(view, max_count=40, max_total=int(big[:, :K].sum(axis=1).max()))
            st.check()
            ref = oracle.append_rows(big[:, :K].astype(npdt))
            _assert_uv_equal(st.finalize(), ref)
            np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12)


def test_empty_and_zero_rows(oracle):
    dev = _dev()
    st = dev.DeviceStats(4)
    st.add_dense(torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
    st.add_dense(torch.zeros((10, 4), dtype=torch.int32, device="cuda"))
    info = st.check()
    assert info["rows"] == 10
    U, V = st.finalize().to_lists()
    assert U == [[], [], [], []] and V == []
    assert st.init_acc.cpu().tolist() == [0.0] * 4


def test_bounds_and_negative_are_reported():
    dev = _dev()
    X = torch.tensor([[1, 2], [300, 4]], dtype=torch.int32, device="cuda")
    st = dev.DeviceStats(2)
    st.add_dense(X, max_count=10, max_total=1000)
    with pytest.raises(ValueError, match="incomplete"):
        st.check()
    st = dev.DeviceStats(2)
    st.add_dense(X, max_count=300, max_total=5)
    with pytest.raises(ValueError, match="incomplete"):
        st.check()
    st = dev.DeviceStats(2)
    st.add_dense(torch.tensor([[1, -2]], dtype=torch.int32, device="cuda"), max_count=5, max_total=5)
    with pytest.raises(ValueError, match="negative"):
        st.check()


def test_chunks_and_logical_shards_add_exactly(oracle):
    """P logical row shards summed == the unsharded build, integer-exact, and the
    same after accumulating chunk by chunk into one object."""
    dev = _dev()
    fx = load_golden("dirmult_k1000")
    X = fx["X"]
    from bayespy_amd.parallel import shard_rows
    whole, uv_whole, _ = _build(X, "u16")
    for P in (2, 3, 8):
        acc = dev.DeviceStats(X.shape[1])
        chunked = dev.DeviceStats(X.shape[1])
        for r in range(P):
            lo, hi = shard_rows(X.shape[0], r, P)
            part = dev.DeviceStats(X.shape[1])
            part.add_dense(_to_dev(X[lo:hi], "u16"))
            acc.merge_(part)
            chunked.add_dense(_to_dev(X[lo:hi], "u16"))
        for st in (acc, chunked):
            st.check()
            U, V = st.finalize().to_lists()
            assert U == fx["U"] and V == fx["V"]
            np.testing.assert_allclose(st.init_acc.cpu().numpy(), whole.init_acc.cpu().numpy(),
                                       rtol=1e-13)


def test_reproducible_bits():
    """Same input, same launch shape -> bit-identical fp64 accumulators (fixed summation order)."""
    fx = load_golden("dirmult_k1000")
    a, _, _ = _build(fx["X"], "u16")
    b, _, _ = _build(fx["X"], "u16")
    assert torch.equal(a.init_acc, b.init_acc)
    assert torch.equal(a.hist, b.hist)


def test_csr_matches_dense(oracle):
    dev = _dev()
    for name in ("dirmult_k50", "dirmult_k1000", "heavy_k20"):
        fx = load_golden(name)
        X = fx["X"].astype(np.int32)
        nz = X != 0
        rowptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))]).astype(np.int64)
        colidx = np.nonzero(nz)[1].astype(np.int32)
        vals = X[nz].astype(np.int32)
        st = dev.DeviceStats(fx["K"])
        st.add_csr(torch.from_numpy(rowptr).cuda(), torch.from_numpy(colidx).cuda(),
                   torch.from_numpy(vals).cuda())
        st.check()
        U, V = st.finalize().to_lists()
        assert U == fx["U"] and V == fx["V"]
        np.testing.assert_allclose(st.init_acc.cpu().numpy(), fx["init_acc"], rtol=1e-12)


def test_large_synthetic_properties(oracle):
    """Size-independent checks at BASELINE's full size (configs[2]: 10M x 1000 uint16 = 20 GB), which
    the oracle cannot reach:
    sum_n U[k][n] = column sum, sum_n V[n] = grand total, V[0] = #non-empty rows,
    U non-increasing, and two half-passes add up to the full pass.  Then the full-size SOLVE against
    the oracle: the solver's cost does not depend on N, so the oracle runs findDirichletPriors on
    the (integer-exact) full-size U/V in milliseconds: alphas 1e-9, loss 1e-12, equal iteration and
    evaluation counts (BASELINE north_star's bars at the metric's own size)."""
    dev = _dev()
    K, N = 1000, 10_000_000
    rng = np.random.default_rng(0)
    alpha = 0.01 + rng.gamma(2.0, 0.05, size=K)
    totals = torch.randint(1, 501, (N,), dtype=torch.int32, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(1))
    X = dev.synth_dirmult(N, K, alpha, totals, seed=20260003)
    for lo in range(0, N, 1_000_000):  # the generator itself: every row has its total
        rowsum = X[lo:lo + 1_000_000].view(torch.int16).to(torch.int32).bitwise_and(0xFFFF).sum(dim=1)
        assert torch.equal(rowsum, totals[lo:lo + 1_000_000])
    st = dev.DeviceStats(K)
    st.add_dense(X, max_count=500, max_total=500)
    info = st.check()
    assert info["rows"] == N
    uv = st.finalize(ragged=False)
    U = uv.Ui.view(K, uv.ldU)
    colsum = torch.zeros(K, dtype=torch.int64, device="cuda")
    for lo in range(0, N, 500_000):
        colsum += X[lo:lo + 500_000].view(torch.int16).to(torch.int64).bitwise_and(0xFFFF).sum(dim=0)
    assert torch.equal(U.sum(dim=1), colsum)
    assert int(uv.Vi.sum()) == int(totals.to(torch.int64).sum())
    assert int(uv.Vi[0]) == N
    assert bool((U[:, 1:] <= U[:, :-1]).all())
    half = dev.DeviceStats(K)
    half.add_dense(X[:N // 2], max_count=500, max_total=500)
    half.add_dense(X[N // 2:], max_count=500, max_total=500)
    half.check()
    assert torch.equal(half.hist, st.hist) and torch.equal(half.histV, st.histV)
    np.testing.assert_allclose(half.init_acc.cpu().numpy(), st.init_acc.cpu().numpy(), rtol=1e-12)
    # ---- full-size solver parity (DE/dme.py:126-190 through the oracle on the same U/V) ----------
    a0 = st.cli_init_alphas()
    out = uv.solve(a0, 50, 0.0, trace_alphas=False)
    Ul, Vl = uv.to_lists()
    full = oracle.stats_from_lists(Ul, Vl)
    ref = oracle.find_priors(full, a0.cpu().numpy(), 50, 0.0)
    np.testing.assert_allclose(out["alpha"], ref["alpha"], rtol=1e-9)
    np.testing.assert_allclose(out["final_loss"], oracle.total_loss(ref["alpha"], full, 0.0), rtol=1e-12)
    assert {1: "A", 2: "B", 3: "C"}[out["exit"]] == ref["exit"]
    # a decision taken on a loss difference of a few ulp (SURVEY 7 "parity-fragile") may legitimately
    # go the other way in the reference's sequential sums: the counts are compared when none was
    if out["fragile_decisions"] == 0:
        assert out["n_iters"] == ref["n_iters"] and out["n_evals"] == ref["n_evals"]


WIDE = [
    # (N, K, ld, dtype): rows split over teams of 2 / 4 / 8 waves, every J the teams are built for,
    # the team edges, odd pitches and K that is no multiple of the vector (unaligned vector loads, the
    # last vector of a row loaded shifted)
    (700, 4104, 4104, "u16"), (300, 5120, 5120, "u16"), (300, 6000, 6000, "u16"),
    (300, 7168, 7176, "u16"), (260, 8192, 8192, "u16"), (150, 8200, 8200, "u16"),
    (150, 12000, 12000, "u16"), (120, 16384, 16384, "u16"), (90, 16392, 16392, "u16"),
    (80, 24000, 24000, "u16"), (70, 32768, 32768, "u16"),
    (500, 2056, 2056, "i32"), (500, 3000, 3000, "i32"), (400, 4096, 4096, "i32"),
    (300, 4104, 4104, "i32"), (200, 6000, 6008, "i32"), (150, 8192, 8192, "i32"),
    (100, 12000, 12000, "i32"), (90, 16384, 16384, "i32"),
    (300, 1001, 1001, "u16"), (300, 4097, 4099, "u16"), (200, 6001, 6001, "u16"),
    (9000, 999, 999, "u16"), (300, 1001, 1001, "i32"), (200, 3001, 3003, "i32"),
    (150, 5000, 5002, "i32"),
]


@pytest.mark.parametrize("N,K,ld,dtype", WIDE)
def test_wide_and_odd_rows_bit_exact_vs_oracle(oracle, N, K, ld, dtype):
    """Rows wider than one wave takes (teams of waves share a row), also at odd pitches and widths:
    same statistics as the oracle, through the dense C entry alone."""
    dev = _dev()
    rng = np.random.default_rng(K * 7 + N)
    npdt = np.uint16 if dtype == "u16" else np.int32
    X = rng.integers(1, 30, size=(N, ld)) * (rng.random((N, ld)) < 0.03)
    # counts beyond the LDS bins (few of those with rows this wide), in the first and last columns
    X[rng.integers(0, N, 20), rng.integers(0, K, 20)] = rng.integers(100, 60000, 20)
    X[0, 0], X[N - 1, K - 1], X[N // 2, K - 1] = 77, 78, 1
    if dtype == "i32":
        X[1, K // 2] = 70000
    Xd = torch.from_numpy(X.astype(npdt)).cuda()[:, :K]
    ref = oracle.append_rows(X[:, :K].astype(npdt))
    st = dev.DeviceStats(K)
    st.add_dense(Xd)
    info = st.check()
    assert info["rows"] == N
    assert info["max_count"] == int(X[:, :K].max())
    assert info["max_total"] == int(X[:, :K].sum(axis=1).max())
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("seed", range(48))
def test_dense_layouts_fuzz_vs_oracle(oracle, seed):
    """Random shapes through the dense entry: any K up to the team range, any pitch, any element
    offset of the base pointer, both element types - whichever kernel the plan picks (vectors of 1
    to 8 elements, 1 to 8 of them per lane, side-by-side rows, packed int32, teams, odd row ends), the
    statistics are the oracle's."""
    dev = _dev()
    rng = np.random.default_rng(1000 + seed)
    dtype = "u16" if seed % 2 == 0 else "i32"
    kmax = 9000 if dtype == "u16" else 5000
    K = int(rng.choice([rng.integers(1, 70), rng.integers(70, 600), rng.integers(600, kmax)]))
    if seed % 6 == 0:
        K = (K + 7) // 8 * 8  # the aligned layouts too
    pad = int(rng.choice([0, 0, 1, 3, 8, 24]))
    off = int(rng.choice([0, 0, 0, 1, 2, 4]))
    N = int(rng.integers(1, 400))
    ld = K + pad
    density = float(rng.choice([0.02, 0.2, 0.9]))
    big = rng.integers(1, 40, size=(N, ld)) * (rng.random((N, ld)) < density)
    for _ in range(4):
        big[rng.integers(0, N), rng.integers(0, K)] = int(rng.integers(100, 60000))
    npdt = np.uint16 if dtype == "u16" else np.int32
    flat = torch.zeros(N * ld + 16, dtype=torch.uint16 if dtype == "u16" else torch.int32, device="cuda")
    flat[off:off + N * ld] = torch.from_numpy(big.astype(npdt).reshape(-1)).cuda()
    view = flat[off:off + N * ld].view(N, ld)[:, :K]
    st = dev.DeviceStats(K)
    st.add_dense(view)
    info = st.check()
    X = big[:, :K]
    assert info == {"rows": N, "max_count": int(X.max()), "max_total": int(X.sum(axis=1).max())}
    ref = oracle.append_rows(X.astype(npdt))
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)


def test_rows_that_end_inside_a_vector_at_size(oracle):
    """K = 1001 uint16 at pitch 1001: no row but the first is even 4-byte aligned and every row ends
    one element into its last vector (loaded shifted, from inside the row); the workspace is
    dme_stats_workspace_bytes alone."""
    from bayespy_amd import _lib
    dev = _dev()
    rng = np.random.default_rng(31)
    N, K = 9000, 1001
    X = (rng.integers(1, 9, size=(N, K)) * (rng.random((N, K)) < 0.05)).astype(np.uint16)
    Xd = torch.from_numpy(X).cuda()
    st = dev.DeviceStats(K, 8, int(X.sum(axis=1).max()))
    L = _lib.lib()
    need = L.dme_stats_workspace_bytes(N, K) + 256
    ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    _lib.check(L.dme_stats_dense_u16(Xd.data_ptr(), N, K, K, st.hist.data_ptr(), st.nbins,
                                     st.histV.data_ptr(), st.tbins, st.init_acc.data_ptr(),
                                     st.status.data_ptr(), 0, ws.data_ptr() + off, need,
                                     torch.cuda.current_stream().cuda_stream))
    assert st.check()["rows"] == N
    ref = oracle.append_rows(X)
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("K,dtype", [(40000, "u16"), (20000, "i32"), (50000, "i32")])
def test_rows_beyond_the_dense_range_go_through_csr(oracle, K, dtype):
    """Dense rows of more columns than the dense kernels take are compressed to CSR by HIP kernels
    (no torch compute) and counted by the CSR entry: same statistics as the oracle.  The C entry
    itself refuses them (DME_ERR_UNSUPPORTED)."""
    from bayespy_amd import _lib
    dev = _dev()
    rng = np.random.default_rng(K)
    N = 300
    X = rng.integers(1, 30, size=(N, K)) * (rng.random((N, K)) < 0.004)
    X[0, 0], X[N - 1, K - 1], X[7] = 3, 4, 0
    X[5, 100] = 500
    npdt = np.uint16 if dtype == "u16" else np.int32
    Xd = torch.from_numpy(X.astype(npdt)).cuda()
    st = dev.DeviceStats(K)
    L = _lib.lib()
    fn = L.dme_stats_dense_u16 if dtype == "u16" else L.dme_stats_dense_i32
    ws = torch.empty(L.dme_stats_workspace_bytes(N, K) + 256, dtype=torch.uint8, device="cuda")
    rc = fn(Xd.data_ptr(), N, K, K, st.hist.data_ptr(), st.nbins, st.histV.data_ptr(), st.tbins,
            st.init_acc.data_ptr(), st.status.data_ptr(), 0, ws.data_ptr() + (-ws.data_ptr()) % 256,
            ws.numel() - 256, torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.ERR_UNSUPPORTED
    st.add_dense(Xd)
    info = st.check()
    assert info["rows"] == N and info["max_count"] == 500
    ref = oracle.append_rows(X.astype(npdt))
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-11, atol=1e-300)
    # the conversion itself: row pointers, ascending columns, values
    counts = torch.empty(N, dtype=torch.int64, device="cuda")
    rowptr = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    nz = L.dme_dense_row_nnz_u16 if dtype == "u16" else L.dme_dense_row_nnz_i32
    cv = L.dme_dense_to_csr_u16 if dtype == "u16" else L.dme_dense_to_csr_i32
    _lib.check(nz(Xd.data_ptr(), N, K, K, counts.data_ptr(), s))
    _lib.check(L.dme_shard_offsets(counts.data_ptr(), N, rowptr.data_ptr(), s))
    nnz = int(rowptr[N].cpu())
    colidx = torch.empty(nnz, dtype=torch.int32, device="cuda")
    vals = torch.empty(nnz, dtype=torch.int32, device="cuda")
    _lib.check(cv(Xd.data_ptr(), N, K, K, rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), s))
    rr, cc = np.nonzero(X)
    assert nnz == len(rr)
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.count_nonzero(X, axis=1))]))
    assert np.array_equal(colidx.cpu().numpy(), cc) and np.array_equal(vals.cpu().numpy(), X[rr, cc])


def test_module_surface_takes_rows_of_a_wide_vocabulary():
    """The drop-in's appendRow with K beyond the dense kernels' range (BASELINE configs[3] has
    K = 50 000): rows are buffered dense and reach the GPU through the CSR conversion."""
    import bayespy_amd.dirichletMultinomialEstimation as DME
    rng = np.random.default_rng(77)
    K = 20000
    data = DME.CompressedRowData(K)
    rows = []
    for _ in range(40):
        row = np.zeros(K, dtype=np.int64)
        row[rng.integers(0, K, 30)] = rng.integers(1, 5, 30)
        rows.append(row)
        data.appendRow(row.tolist(), 1)
    X = np.stack(rows)
    assert [int(v) for v in data.V] == [int((X.sum(axis=1) > n).sum()) for n in range(int(X.sum(axis=1).max()))]
    k = int(np.argmax(X.max(axis=0)))
    assert [int(u) for u in data.U[k]] == [int((X[:, k] > n).sum()) for n in range(int(X[:, k].max()))]


def _csr_direct(dev, rowptr, colidx, vals, K, nbins, tbins, partitioned):
    """dme_stats_csr_i32 through the C ABI with or without the partition workspace."""
    from bayespy_amd import _lib
    L = _lib.lib()
    N, nnz = rowptr.numel() - 1, colidx.numel()
    hist = torch.zeros(K * nbins, dtype=torch.int64, device="cuda")
    histV = torch.zeros(tbins, dtype=torch.int64, device="cuda")
    acc = torch.zeros(K, dtype=torch.float64, device="cuda")
    status = torch.zeros(8, dtype=torch.int64, device="cuda")
    ws, wsb, keep = 0, 0, None
    if partitioned:
        need = L.dme_stats_csr_workspace_bytes(N, nnz, K)
        assert need > 0
        keep = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
        off = (-keep.data_ptr()) % 256
        ws, wsb = keep.data_ptr() + off, need
    _lib.check(L.dme_stats_csr_i32(rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), N, nnz, K,
                                   hist.data_ptr(), nbins, histV.data_ptr(), tbins, acc.data_ptr(),
                                   status.data_ptr(), 0, ws, wsb, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return hist, histV, acc, status


@pytest.mark.parametrize("K", [700, 50_000, 600_000])
def test_csr_partitioned_equals_atomics_path(K):
    """The bucket-partitioned CSR pass against the plain atomics kernel on hostile input: empty
    rows, zero and negative counts, column indices out of range, counts beyond the LDS bins, a hot
    column, row totals beyond the LDS total bins and beyond tbins, counts beyond nbins."""
    dev = _dev()
    rng = np.random.default_rng(K)
    N = 30_000
    lens = rng.integers(0, 40, size=N)
    lens[rng.integers(0, N, size=50)] = 0
    lens[7] = 3000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = (rng.zipf(1.3, size=nnz) - 1) % K
    col[rng.random(nnz) < 0.3] = 5                      # hot column
    val = rng.geometric(0.5, size=nnz)
    big = rng.random(nnz) < 0.02
    val[big] = rng.integers(61, 400, size=int(big.sum()))    # beyond BL
    val[rng.random(nnz) < 0.01] = 0
    val[rng.integers(0, nnz, size=5)] = -3
    val[rng.integers(0, nnz, size=5)] = 5000            # >= nbins
    col[rng.integers(0, nnz, size=5)] = K + 17
    col[rng.integers(0, nnz, size=5)] = -1
    nbins, tbins = 450, 20_000                          # row 7 exceeds tbins; others beyond 8192? one:
    p0 = rowptr[11]
    if lens[11] > 0:
        val[p0] = 399
    rp = torch.from_numpy(rowptr).cuda()
    ci = torch.from_numpy(col.astype(np.int32)).cuda()
    vv = torch.from_numpy(val.astype(np.int32)).cuda()
    a = _csr_direct(dev, rp, ci, vv, K, nbins, tbins, partitioned=False)
    b = _csr_direct(dev, rp, ci, vv, K, nbins, tbins, partitioned=True)
    assert torch.equal(a[0], b[0]), "hist"
    assert torch.equal(a[1], b[1]), "histV"
    assert torch.equal(a[3], b[3]), (a[3].tolist(), b[3].tolist())
    st = a[3].tolist()
    assert st[0] == N and st[1] > 0 and st[3] > 0
    np.testing.assert_allclose(b[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-12, atol=1e-300)


def test_csr_partitioned_row_totals_beyond_lds_bins(oracle):
    """Clean rows with totals up to 30 000 (LDS holds the first 8192 total bins) against the oracle."""
    dev = _dev()
    rng = np.random.default_rng(77)
    N, K = 5000, 3000
    lens = rng.integers(1, 30, size=N)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = np.concatenate([rng.choice(K, size=l, replace=False) for l in lens]).astype(np.int32)
    val = rng.integers(1, 8, size=nnz).astype(np.int32)
    heavy = rng.random(nnz) < 0.01
    val[heavy] = rng.integers(1000, 3000, size=int(heavy.sum()))
    st = dev.DeviceStats(K)
    st.add_csr(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda())
    info = st.check()
    assert info["rows"] == N and info["max_total"] > 8192
    ref = oracle.append_rows_csr(rowptr, col, val, K)
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-11, atol=1e-300)


def test_csr_rowptr_beyond_nnz_is_refused():
    """rowptr that spans more entries than the stated nnz (the partition workspace is sized by nnz):
    rows beyond it are not read and are flagged; check() raises."""
    dev = _dev()
    rng = np.random.default_rng(3)
    N, K, per = 4000, 3000, 20
    nnz = N * per
    rowptr = torch.arange(0, nnz + 1, per, dtype=torch.int64, device="cuda")
    col = torch.from_numpy(rng.integers(0, K, size=nnz).astype(np.int32)).cuda()
    val = torch.ones(nnz, dtype=torch.int32, device="cuda")
    hist, histV, acc, status = _csr_direct_nnz(rowptr, col, val, K, 4, 64, claimed_nnz=nnz // 2)
    assert int(status[6]) >= 1
    assert int(histV.sum()) == int(status[0]) <= N // 2  # nothing beyond the stated entries was read
    st = dev.DeviceStats(K, 3, 63)
    st.status += status
    with pytest.raises(ValueError, match="spans more entries"):
        st.check()


def test_csr_rowptr_running_backwards_is_refused_on_the_partitioned_path():
    """A non-monotonic rowptr (rows that run backwards) whose last entry still equals nnz: the flat
    count pass and the row-wise scatter pass would disagree about the entries of a slab.  The count
    pass checks every row's pointers before it reads the row; such rows are skipped and flagged,
    check() raises."""
    dev = _dev()
    rng = np.random.default_rng(4)
    N, K, per = 4000, 3000, 20
    nnz = N * per
    rp = np.arange(0, nnz + 1, per, dtype=np.int64)
    rp[1000], rp[1001] = rp[1001], rp[1000]  # ..., 20020, 20000, ...: decreasing once
    rp[3000] = -5                            # and a negative one
    rowptr = torch.from_numpy(rp).cuda()
    col = torch.from_numpy(rng.integers(0, K, size=nnz).astype(np.int32)).cuda()
    val = torch.ones(nnz, dtype=torch.int32, device="cuda")
    hist, histV, acc, status = _csr_direct_nnz(rowptr, col, val, K, 4, 64, claimed_nnz=nnz)
    assert int(status[6]) >= 1
    assert int(histV.sum()) == int(status[0]) < N
    st = dev.DeviceStats(K, 3, 63)
    st.status += status
    with pytest.raises(ValueError, match="rowptr"):
        st.check()


def test_csr_small_path_skips_rows_of_a_malformed_rowptr():
    """The atomics kernel (small inputs, no workspace) is told nnz and does not touch a row whose
    pointers are negative, decreasing or beyond nnz; check() raises."""
    dev = _dev()
    K = 50
    rowptr = torch.tensor([0, 3, 2, 6, 40], dtype=torch.int64, device="cuda")  # row 1 backwards, row 3 beyond nnz
    col = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8], dtype=torch.int32, device="cuda")
    val = torch.ones(8, dtype=torch.int32, device="cuda")
    hist, histV, acc, status = _csr_direct(dev, rowptr, col, val, K, 4, 16, partitioned=False)
    st = status.cpu().tolist()
    assert st[6] == 2 and st[0] == 2           # two rows refused, two counted
    assert int(histV[3]) == 1 and int(histV[4]) == 1 and int(histV.sum()) == 2
    s2 = dev.DeviceStats(K, 3, 15)
    s2.status += status
    with pytest.raises(ValueError, match="rowptr"):
        s2.check()


def _csr_direct_nnz(rowptr, colidx, vals, K, nbins, tbins, claimed_nnz):
    from bayespy_amd import _lib
    L = _lib.lib()
    N = rowptr.numel() - 1
    hist = torch.zeros(K * nbins, dtype=torch.int64, device="cuda")
    histV = torch.zeros(tbins, dtype=torch.int64, device="cuda")
    acc = torch.zeros(K, dtype=torch.float64, device="cuda")
    status = torch.zeros(8, dtype=torch.int64, device="cuda")
    need = L.dme_stats_csr_workspace_bytes(N, claimed_nnz, K)
    assert need > 0
    keep = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    off = (-keep.data_ptr()) % 256
    _lib.check(L.dme_stats_csr_i32(rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), N, claimed_nnz, K,
                                   hist.data_ptr(), nbins, histV.data_ptr(), tbins, acc.data_ptr(),
                                   status.data_ptr(), 0, keep.data_ptr() + off, need,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return hist, histV, acc, status


def test_csr_without_entries():
    """CSR rows that store nothing: rows are counted, no statistics change."""
    dev = _dev()
    st = dev.DeviceStats(7)
    rowptr = torch.zeros(6, dtype=torch.int64, device="cuda")
    empty = torch.zeros(0, dtype=torch.int32, device="cuda")
    st.add_csr(rowptr, empty, empty)
    info = st.check()
    assert info["rows"] == 5 and info["max_count"] == 0 and info["max_total"] == 0
    assert int(st.hist.sum()) == 0 and int(st.histV.sum()) == 0
    U, V = st.finalize().to_lists()
    assert U == [[] for _ in range(7)] and V == []


def test_packed_int32_kernel_exceptions(oracle):
    """int32 rows of 8-column chunks take the packed kernel (counts travel as 16-bit pairs): counts
    of 65536 and more are settled one by one and must land in the same bins; negatives are reported."""
    dev = _dev()
    rng = np.random.default_rng(42)
    # (10, 100, 600, 1001, 3000, 4096: the kernels that refill their buffers early and read such cells again)
    for K in (512, 1000, 2048, 4096, 10, 100, 600, 1001, 3000):
        N = 300
        X = rng.integers(1, 40, size=(N, K)) * (rng.random((N, K)) < max(0.05, 2.0 / K))
        for v in (65535, 65536, 70000, 200000, 65537):
            X[rng.integers(0, N), rng.integers(0, K)] = v
        ref = oracle.append_rows(X.astype(np.int32))
        st, uv, info = _build(X, "i32")
        assert info["max_count"] == 200000
        _assert_uv_equal(uv, ref)
        np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-12, atol=1e-300)
        Xn = X.copy()
        Xn[5, 17 % K] = -3
        bad = dev.DeviceStats(K)
        bad.add_dense(_to_dev(Xn, "i32"), max_count=200000, max_total=int(X.sum(axis=1).max()))
        with pytest.raises(ValueError, match="negative"):
            bad.check()


@pytest.mark.parametrize("seed", range(12))
def test_csr_partitioned_fuzz_vs_oracle(oracle, seed):
    """Random CSR shapes through the partitioned pass against the oracle: K from one bucket to
    thousands of buckets, empty rows, rows longer than a scatter batch, duplicate-free columns."""
    dev = _dev()
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([37, 256, 257, 5000, 70_000, 300_000]))
    N = int(rng.integers(2000, 9000))
    mean_len = int(rng.choice([3, 15, 40]))
    lens = rng.poisson(mean_len, size=N)
    lens[rng.random(N) < 0.1] = 0
    if K >= 5000:
        lens[int(rng.integers(0, N))] = 3000          # one row longer than a 2048-entry batch
    lens = np.minimum(lens, K)
    while lens.sum() < 40_000:                          # stay on the partitioned path
        lens = np.minimum(lens * 2 + 1, K)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([rng.choice(K, size=l, replace=False) if l else np.zeros(0, dtype=np.int64)
                          for l in lens]).astype(np.int32)
    val = rng.geometric(0.4, size=col.size).astype(np.int32)
    val[rng.random(col.size) < 0.01] = rng.integers(60, 500)
    val[rng.random(col.size) < 0.02] = 0
    st = dev.DeviceStats(K)
    st.add_csr(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda())
    assert st.check()["rows"] == N
    ref = oracle.append_rows_csr(rowptr, col, val, K)
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("K,huge", [(50_000, False), (50_000, True), (65_536, False), (5000, False)])
def test_csr_rows_longer_than_a_tile_vs_oracle(oracle, K, huge):
    """Rows of 4097 ... 20 000 stored entries through the tile partition kernel (4096 entries per tile,
    a workgroup's range ~ a few tiles): a row that fills a tile exactly, rows that span two, three and
    five tiles and two workgroups' ranges, the first and the last row of the matrix among them, long rows
    back to back, empty rows around them; `huge` plants a count of 2^19 in a long row (64-bit row
    totals).  Row totals of the longest rows exceed what a record holds (16 383): their entries take the
    rare path.  DE/dme.py:113-123 against the oracle, cell by cell."""
    dev = _dev()
    rng = np.random.default_rng(4242 + K + int(huge))
    N = 3000
    lens = rng.poisson(12, size=N)
    lens[rng.random(N) < 0.1] = 0
    for j, l in [(0, 4500), (100, 4096), (101, 4097), (102, 4095), (700, 9000), (701, 0), (702, 8200),
                 (1500, 20_000), (2200, 12_288), (2201, 5000), (N - 1, 6000)]:
        lens[j] = min(l, K)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([rng.choice(K, size=l, replace=False) if l else np.zeros(0, dtype=np.int64)
                          for l in lens]).astype(np.int32)
    val = rng.geometric(0.6, size=col.size).astype(np.int32)
    val[rng.random(col.size) < 0.02] = 0
    val[rng.random(col.size) < 0.002] = 300            # beyond a record's 8-bit count
    if huge:
        val[rowptr[700] + 5000] = 1 << 19
        val[rowptr[20] : rowptr[21]] = 1               # a short row in the same run of tiles stays exact
    st = dev.DeviceStats(K)
    st.add_csr(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda())
    info = st.check()
    assert info["rows"] == N
    totals = np.add.reduceat(np.append(val, 0).astype(np.int64), rowptr[:-1])
    totals[lens == 0] = 0
    assert info["max_total"] == int(totals.max())
    assert K < 20_000 or info["max_total"] >= (1 << 14)  # (K = 5000 caps the rows at 5000 entries)
    ref = oracle.append_rows_csr(rowptr, col, val, K)
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("K,N,max_total", [(512, 16_000_000, 300), (8, 40_000_000, 40),
                                           (24, 5_000_000, 50), (2, 6_000_000, 60)])
def test_more_rows_than_one_round_of_workgroups(K, N, max_total):
    """Beyond 256 x 60000 rows the grid grows to several workgroups per CU (a workgroup's 16-bit LDS
    counters hold 65535 rows): the interleaved row mapping must still cover every row once - row
    count, column sums, grand total and the row-total histogram against torch reductions; K = 8 is
    the side-by-side layout (64 rows per wave); K = 24 carries 21 rows per wave (a bin word per
    lane, 16 copies of the row-total bins), K = 2 one row per lane."""
    dev = _dev()
    rng = np.random.default_rng(K)
    alpha = 0.05 + rng.gamma(2.0, 0.3, size=K)
    totals = torch.randint(1, max_total + 1, (N,), dtype=torch.int32, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(K))
    X = dev.synth_dirmult(N, K, alpha, totals, seed=77)
    st = dev.DeviceStats(K)
    st.add_dense(X, max_count=max_total, max_total=max_total)
    assert st.check()["rows"] == N
    uv = st.finalize(ragged=False)
    U = uv.Ui.view(K, uv.ldU)
    colsum = torch.zeros(K, dtype=torch.int64, device="cuda")
    step = max(1, (256 << 20) // K)
    for lo in range(0, N, step):
        colsum += X[lo:lo + step].view(torch.int16).to(torch.int64).bitwise_and(0xFFFF).sum(dim=0)
    assert torch.equal(U.sum(dim=1), colsum)
    assert torch.equal(st.histV[:max_total + 1],
                       torch.bincount(totals.to(torch.int64), minlength=max_total + 1))
    assert int(uv.Vi[0]) == N


@pytest.mark.parametrize("K,N,dtype", [(3000, 400_000, "u16"), (8192, 200_000, "u16"), (16384, 60_000, "u16"),
                                       (32768, 40_000, "u16"), (2048, 300_000, "i32"), (4096, 200_000, "i32"),
                                       (8192, 80_000, "i32"), (16384, 40_000, "i32"), (6001, 90_000, "u16")])
def test_long_rows_at_size_against_torch_reductions(K, N, dtype):
    """Rows shared by teams of waves (and K = 3000: six vectors per lane; K = 6001: odd row ends in a
    team) over enough rows that every workgroup runs many loop steps of the interleaved / slab
    mappings: row count, column sums, grand total and the row-total histogram against torch
    reductions (size-independent properties; the small shapes are compared with the oracle cell by
    cell in test_wide_and_odd_rows_bit_exact_vs_oracle)."""
    dev = _dev()
    rng = np.random.default_rng(K)
    alpha = 0.002 + rng.gamma(2.0, 0.01, size=K)
    max_total = 400
    totals = torch.randint(1, max_total + 1, (N,), dtype=torch.int32, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(K))
    X = dev.synth_dirmult(N, K, alpha, totals, seed=78,
                          dtype=torch.uint16 if dtype == "u16" else torch.int32)
    st = dev.DeviceStats(K)
    st.add_dense(X, max_count=max_total, max_total=max_total)
    assert st.check()["rows"] == N
    uv = st.finalize(ragged=False)
    U = uv.Ui.view(K, uv.ldU)
    colsum = torch.zeros(K, dtype=torch.int64, device="cuda")
    step = max(1, (64 << 20) // K)
    for lo in range(0, N, step):
        blk = X[lo:lo + step]
        if dtype == "u16":
            blk = blk.view(torch.int16).to(torch.int64).bitwise_and(0xFFFF)
        colsum += blk.to(torch.int64).sum(dim=0)
    assert torch.equal(U.sum(dim=1), colsum)
    assert int(colsum.sum()) == int(totals.to(torch.int64).sum())
    assert torch.equal(st.histV[:max_total + 1],
                       torch.bincount(totals.to(torch.int64), minlength=max_total + 1))
    assert int(uv.Vi[0]) == N


@pytest.mark.parametrize("K,dtype,big", [(16, "u16", 0), (2, "i32", 0), (24, "u16", 0), (16, "u16", 700),
                                         (8, "u16", 500), (2, "i32", 100_000)])
def test_per_lane_bins_bit_exact_vs_oracle(oracle, K, dtype, big):
    """From 4 M rows on, rows that lie 16 or more side by side in a wave keep several copies of a
    column's bin words (one per lane when the count bins leave room, fewer when they would not fit)
    and a copy of the row-total bins per row slot (summed by hist_reduce_kernel): U/V bit-exact and
    the mean accumulator to 1e-12 against the oracle's loops at that size; one row total dominates
    (the fixed-number-of-trials case the layout is for); `big`: some counts far beyond the rest
    (the copies are halved until every bin is in LDS, or, at 100 000, the largest go to global
    atomics)."""
    N = 4_200_000
    rng = np.random.default_rng(K)
    X = rng.integers(0, 9, size=(N, K), dtype=np.int32)
    X[::3] = 0
    X[::3, 0] = 60                                        # a third of the rows: total exactly 60
    if big:
        X[rng.integers(0, N, 5000), rng.integers(0, K, 5000)] = rng.integers(61, big + 1, 5000)
    ref = oracle.append_rows(X.astype(np.uint16) if dtype == "u16" else X)
    st, uv, info = _build(X, dtype)
    assert info["rows"] == N
    _assert_uv_equal(uv, ref)
    # the mean accumulator: the oracle adds 4.2 M terms one after the other as the reference does
    # (rounding ~1e-11 relative at this length), the kernel in a tree; both against an extended-
    # precision sum, the kernel at the tighter bound
    n = X.sum(axis=1, dtype=np.int64)
    exact = np.zeros(K, dtype=np.longdouble)
    for lo in range(0, N, 500_000):
        w = np.longdouble(1) / (1 + n[lo:lo + 500_000]).astype(np.longdouble)
        exact += (X[lo:lo + 500_000].astype(np.longdouble) * w[:, None]).sum(axis=0)
    exact = exact.astype(np.float64)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), exact, rtol=1e-13)
    np.testing.assert_allclose(ref.init_acc, exact, rtol=1e-10)


@pytest.mark.parametrize("K,N,big,cus", [(2, 1_000_003, 0, 0), (4, 700_001, 0, 0), (2, 4_200_001, 300, 0),
                                         (4, 300_000, 60000, 0), (2, 65_536, 0, 0), (8, 500_000, 0, 4),
                                         (16, 400_000, 200, 4), (24, 300_001, 0, 4), (104, 300_000, 0, 4),
                                         (8, 4_100_000, 900, 16),
                                         (104, 50_000, 0, 0), (200, 100_000, 300, 0), (264, 60_000, 0, 0),
                                         (520, 50_000, 0, 0), (600, 40_000, 300, 0), (1016, 20_000, 0, 0),
                                         (200, 150_000, 0, 2), (48, 4_100_000, 0, 0), (400, 50_000, 0, 0),
                                         (3000, 9_000, 300, 0), (4096, 7_000, 0, 0), (8000, 5_000, 0, 0)])
def test_narrow_rows_with_32_bit_counters_bit_exact_vs_oracle(oracle, K, N, big, cus):
    """uint16 rows of 2 or 4 columns that lie back to back are read as 16-byte vectors of 4 or 2
    whole rows (the rows that do not fill a last vector go through the ordinary kernel afterwards),
    and side-by-side rows that would take more than one round of workgroups count in 32-bit LDS
    words, one workgroup per CU (`cus`: the library is told it has that few CUs, so that a few
    hundred thousand rows are already more than a round): U/V bit-exact and the mean accumulator to
    1e-12 against the oracle's loops; `big`: a few counts far beyond the rest (copies of the bin
    words are halved until every bin fits; at 60000 they go to global atomics).  K = 104 ... 600:
    rows whose vectors do not fill a wave's lanes on their own lie side by side with 2 to 4 vectors
    per lane (K = 200: 9 lanes x 3 vectors, 7 rows per wave; K = 48 at 4.1 M rows: 2 lanes x 3
    vectors, 32 rows per wave with a bin word per row slot; K = 400: 10 lanes x 5 vectors).
    K = 3000, 4096, 8000: teams of two or four waves with 3 or 4 vectors per lane."""
    _narrow_case(oracle, K, N, big, cus, "u16")


@pytest.mark.parametrize("K,N,big,cus", [(8, 500_000, 0, 4), (16, 400_000, 200, 4), (104, 300_000, 0, 4),
                                         (24, 300_001, 70000, 4), (300, 200_000, 0, 2), (100, 60_000, 0, 0),
                                         (10, 500_000, 0, 4)])
def test_narrow_int32_rows_bit_exact_vs_oracle(oracle, K, N, big, cus):
    """int32 rows of 8 to 511 columns take the packed kernel side by side in a wave, with 32-bit LDS
    counters when they would need more than one round of workgroups (`cus` as above); `big` = 70000:
    counts beyond 16 bits are settled one by one.  Same bars as for uint16 rows."""
    _narrow_case(oracle, K, N, big, cus, "i32")


def _narrow_case(oracle, K, N, big, cus, dtype):
    from bayespy_amd import _lib
    rng = np.random.default_rng(K * 1000 + N % 1000)
    X = rng.integers(0, 7, size=(N, K), dtype=np.int32)
    X[::5] = 0
    X[::5, K - 1] = 33                                    # a fifth of the rows: total exactly 33
    X[1::7] = 0                                           # empty rows
    if big:
        X[rng.integers(0, N, 3000), rng.integers(0, K, 3000)] = rng.integers(40, big + 1, 3000)
    ref = oracle.append_rows(X.astype(np.uint16) if dtype == "u16" else X)
    L = _lib.lib()
    try:
        if cus:
            _lib.check(L.dme_set_cu_limit(cus))
        st, uv, info = _build(X, dtype)
        torch.cuda.synchronize()
    finally:
        _lib.check(L.dme_set_cu_limit(0))
    assert info["rows"] == N
    _assert_uv_equal(uv, ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-10)
    n = X.sum(axis=1, dtype=np.int64)
    exact = (X.astype(np.longdouble) / (1 + n).astype(np.longdouble)[:, None]).sum(axis=0).astype(np.float64)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), exact, rtol=1e-12)


@pytest.mark.parametrize("dtype", ["u16", "i32", "i32-negatives"])
# (the last two: 81 and 151 bins a side - four copies of the pair table in LDS, and one)
@pytest.mark.parametrize("max_count,max_total", [(40, 80), (20, 30), (33, 35), (80, 120), (150, 200)])
def test_pair_counts_of_two_columns_equal_the_general_kernel(oracle, max_count, max_total, dtype):
    """K = 2 in uint16 rows that lie back to back counts the pairs (x0, x1) in LDS and derives
    hist / histV / the mean accumulator from them (stats_pair_kernel); the same rows at a pitch of 4
    take the general kernel.  Both must leave the same histograms and the same status words - also
    when the caller's bounds are too small (counts of 33 against nbins = 21: those rows are settled
    one by one, status[1] / status[2] report them) - and, within bounds, the oracle's U / V.
    int32 rows go two to a vector; "i32-negatives": some negative counts, which both kernels report
    in status[3] and count as zero."""
    dev = _dev()
    N = 300_003                                            # three rows beyond the last full vector
    rng = np.random.default_rng(max_count)
    X = rng.integers(0, 9, size=(N, 2), dtype=np.int32)
    X[::5] = (0, 33)
    X[1::7] = 0
    X[2::11] = (17, 0)
    if max_count >= 80:                                   # counts over the whole table
        X[3::19] = rng.integers(0, max_count // 2, size=X[3::19].shape)
    negatives = dtype == "i32-negatives"
    if negatives:
        X[3::13, 0] = -5
        X[4::17] = (-1, -2)
    if dtype == "u16":
        flat = torch.from_numpy(X.astype(np.uint16)).cuda()
    else:
        flat = torch.from_numpy(X).cuda()
    wide = torch.zeros((N, 4), dtype=flat.dtype, device="cuda")
    wide[:, :2] = flat
    a = dev.DeviceStats(2, max_count, max_total)
    a.add_dense(flat, max_count=max_count, max_total=max_total)
    b = dev.DeviceStats(2, max_count, max_total)
    b.add_dense(wide[:, :2], max_count=max_count, max_total=max_total)
    torch.cuda.synchronize()
    assert a.nbins == b.nbins and a.tbins == b.tbins
    assert torch.equal(a.hist.view(2, -1)[:, 1:], b.hist.view(2, -1)[:, 1:])
    assert torch.equal(a.histV[1:], b.histV[1:])
    assert torch.equal(a.status, b.status)
    np.testing.assert_allclose(a.init_acc.cpu().numpy(), b.init_acc.cpu().numpy(), rtol=1e-12)
    if negatives:
        assert int(a.status[3]) == int((X < 0).sum())
        with pytest.raises(ValueError, match="negative"):
            a.check()
    elif max_count >= 33 and max_total >= 35:
        info = a.check()
        assert info["rows"] == N
        ref = oracle.append_rows(X.astype(np.uint16) if dtype == "u16" else X)
        _assert_uv_equal(a.finalize(), ref)
        np.testing.assert_allclose(a.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-10)
    else:
        with pytest.raises(ValueError, match="incomplete"):
            a.check()


@pytest.mark.parametrize("dtype", ["u16", "i32"])
@pytest.mark.parametrize("N,K,pad", [(1, 1, 0), (300, 8, 0), (257, 1000, 0), (100, 1000, 24), (90, 13, 3),
                                     (5000, 64, 0)])
def test_bounds_pass(dtype, N, K, pad):
    """dme_bounds_dense_*: largest cell and largest row total (positive parts), vector and scalar
    layouts, padded rows; and add_dense without stated bounds sizes its histograms from it."""
    from bayespy_amd import _lib
    dev = _dev()
    rng = np.random.default_rng(N + K + pad)
    hi = 60000 if dtype == "u16" else 3_000_000
    X = rng.integers(0, 50, size=(N, K + pad)).astype(np.int64)
    X[rng.integers(0, N), rng.integers(0, K)] = hi
    if dtype == "i32" and N > 1:
        X[0, 0] = -7  # ignored here, reported by the statistics pass
    Xd = _to_dev(X, dtype)[:, :K]
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    fn = _lib.lib().dme_bounds_dense_u16 if dtype == "u16" else _lib.lib().dme_bounds_dense_i32
    _lib.check(fn(Xd.data_ptr(), N, K, Xd.stride(0), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    pos = np.maximum(X[:, :K], 0)
    assert out.cpu().tolist() == [int(pos.max()), int(pos.sum(axis=1).max())]
    if not (dtype == "i32" and N > 1):
        st = dev.DeviceStats(K)
        st.add_dense(Xd)
        info = st.check()
        assert info["max_count"] == int(pos.max()) and info["max_total"] == int(pos.sum(axis=1).max())
        # (the dense bins stop at max(64, 2^25 // K); beyond them the overflow store takes over)
        assert st.nbins == min(int(pos.max()) + 1, st.dense_bins)
        assert (st.overflow is not None) == (int(pos.max()) + 1 > st.dense_bins)
        assert st.tbins == int(pos.sum(axis=1).max()) + 1


def test_bounds_pass_csr():
    """dme_bounds_csr_i32 against numpy on ragged rows (empty rows, one long row, a negative value)."""
    from bayespy_amd import _lib
    rng = np.random.default_rng(11)
    N, K = 5000, 300
    lens = rng.integers(0, 40, size=N)
    lens[17] = 9000
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    nnz = int(rowptr[-1])
    vals = rng.integers(0, 9, size=nnz).astype(np.int32)
    vals[5] = -3
    vals[nnz // 2] = 123456
    pos = np.maximum(vals, 0).astype(np.int64)
    tot = np.add.reduceat(np.concatenate([pos, [0]]), rowptr[:-1]) * (lens > 0)
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    rp, vd = torch.from_numpy(rowptr).cuda(), torch.from_numpy(vals).cuda()
    _lib.check(_lib.lib().dme_bounds_csr_i32(rp.data_ptr(), vd.data_ptr(), N, nnz, out.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
    assert out.cpu().tolist() == [int(pos.max()), int(tot.max())]


@pytest.mark.parametrize("K,nhot,dtype", [(1000, 6, "u16"), (1000, 40, "u16"), (1000, 6, "i32"), (16, 3, "u16"),
                                          (200, 8, "i32")])
def test_hot_columns_bit_exact_vs_oracle(oracle, K, nhot, dtype):
    """Counts that routinely exceed the packed LDS bins (a few hot columns hold most of every row, more
    than one cell per row goes to the 64-bit global atomics of the large-count path): U/V bit-exact.
    Throughput on such data: benchmarks/hot_columns.py, profiles/r2_summary.md."""
    dev = _dev()
    rng = np.random.default_rng(K * 100 + nhot)
    N, M = 6000, 500
    alpha = np.full(K, 0.002)
    hot = rng.choice(K, size=min(nhot, K), replace=False)
    alpha[hot] = 0.3  # a row's mass sits in one or two of the hot columns
    tot = rng.integers(1, M + 1, size=N)
    X = np.stack([rng.multinomial(t, rng.dirichlet(alpha)) for t in tot]).astype(np.int32)
    assert X.max() > 200
    st = dev.DeviceStats(K)
    st.add_dense(_to_dev(X, dtype), max_count=M, max_total=M)
    info = st.check()
    assert info["rows"] == N and info["max_count"] == int(X.max())
    ref = oracle.append_rows(X)
    _assert_uv_equal(st.finalize(), ref)
    np.testing.assert_allclose(st.init_acc.cpu().numpy(), ref.init_acc, rtol=1e-11, atol=1e-300)
