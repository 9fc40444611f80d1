"""N > 1 on the device: two processes share the one GPU of the test box (gloo moves the device
tensors; RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench).
Each rank builds the statistics of its row shard with the HIP kernel, DeviceStats.allreduce()
sums them, and both ranks must end with the reference's U/V and the same alphas."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, X, K, out, rows_bound=None, compact=None):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bayespy_amd import device
    from bayespy_amd.parallel import shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_rows(X.shape[0], rank, world)
    st = device.DeviceStats(K)
    st.add_dense(torch.from_numpy(np.ascontiguousarray(X[lo:hi])).cuda())
    if compact:  # extents differ between the ranks; agreed with the columns' extents
        st.allreduce(compact=True, rows_bound=rows_bound)
    elif rows_bound is None:
        st.allreduce()
    else:  # the 32-bit message needs agreed extents: size every rank's histograms alike first
        st.reserve(int(X.max()), int(X.sum(axis=1).max()))
        st.allreduce(same_extents=True, rows_bound=rows_bound)
    info = st.check()
    uv = st.finalize()
    U, V = uv.to_lists()
    sol = uv.solve(st.cli_init_alphas(), 50, 0.0)
    out[rank] = (U, V, info["rows"], sol["alpha"], sol["n_iters"])
    dist.destroy_process_group()


@pytest.mark.parametrize("rows_bound,compact", [(None, None), (2000, None), (None, True), (2000, True)])
def test_two_ranks_allreduce_then_identical_solve(rows_bound, compact):
    import torch.multiprocessing as mp
    from conftest import load_golden
    fx = load_golden("dirmult_k1000")
    X = fx["X"].astype(np.uint16)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, X, fx["K"], out, rows_bound, compact), nprocs=2, join=True)
    ref = [s for s in fx["solves"] if s["variant"] == "DE" and s["init_name"] == "cli"][0]
    for r in range(2):
        U, V, rows, alpha, n_iters = out[r]
        assert U == fx["U"] and V == fx["V"] and rows == fx["N"]
        np.testing.assert_allclose(alpha, ref["final"], rtol=1e-9)
        assert n_iters == ref["n_iters"]
    assert np.array_equal(out[0][3], out[1][3])  # replicated solve: bit-identical on every rank


@pytest.mark.parametrize("world,K,nbins,tbins", [(3, 5, 4, 7), (8, 1000, 33, 65), (2, 1, 2, 2)])
def test_shard_message_kernels_match_the_host_layout(world, K, nbins, tbins):
    """dme_shard_pack / dme_shard_unpack against the message the host code builds for gloo ranks
    (bayespy_amd/parallel.py): every rank's packed message word for word; then, on the summed
    message, the maxima over ranks and the accumulators added in rank order, bit for bit."""
    from bayespy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(world * 1000 + K)
    nh, nv = K * nbins, tbins
    nmsg = nh + nv + 8 + world * (2 + K)
    assert L.dme_shard_message_words(nh, nv, K, world) == nmsg
    stream = torch.cuda.current_stream().cuda_stream
    total = np.zeros(nmsg, dtype=np.int64)
    accs, stats = [], []
    for rank in range(world):
        hist = rng.integers(0, 1 << 40, size=nh, dtype=np.int64)
        histV = rng.integers(0, 1 << 40, size=nv, dtype=np.int64)
        status = rng.integers(0, 1 << 30, size=8, dtype=np.int64)
        acc = rng.random(K) * 10.0 ** rng.integers(-8, 8, size=K)
        accs.append(acc)
        stats.append(status)
        want = np.zeros(nmsg, dtype=np.int64)
        want[:nh], want[nh:nh + nv] = hist, histV
        want[nh + nv:nh + nv + 8] = status
        want[nh + nv + 4:nh + nv + 6] = 0
        slots = want[nh + nv + 8:].reshape(world, 2 + K)
        slots[rank, :2] = status[4:6]
        slots[rank, 2:] = acc.view(np.int64)
        d = [torch.from_numpy(a).cuda() for a in (hist, histV, status, acc)]
        msg = torch.full((nmsg,), -1, dtype=torch.int64, device="cuda")
        _lib.check(L.dme_shard_pack(d[0].data_ptr(), nh, d[1].data_ptr(), nv, d[2].data_ptr(),
                                    d[3].data_ptr(), K, rank, world, msg.data_ptr(), stream))
        assert np.array_equal(msg.cpu().numpy(), want)
        total += want
    msg = torch.from_numpy(total).cuda()
    out = torch.empty(K, dtype=torch.float64, device="cuda")
    _lib.check(L.dme_shard_unpack(msg.data_ptr(), nh, nv, K, world, out.data_ptr(), stream))
    got = msg.cpu().numpy()
    st = got[nh + nv:nh + nv + 8]
    summed = np.sum(stats, axis=0)
    assert np.array_equal(st[:4], summed[:4]) and np.array_equal(st[6:], summed[6:])
    assert st[4] == max(s[4] for s in stats) and st[5] == max(s[5] for s in stats)
    ref = accs[0].copy()
    for r in range(1, world):
        ref = ref + accs[r]
    assert np.array_equal(out.cpu().numpy(), ref)  # rank order, bit for bit
    assert np.array_equal(got[:nh + nv], total[:nh + nv])


def test_shard_pack_rejects_bad_rank():
    from bayespy_amd import _lib
    z = torch.zeros(16, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().dme_shard_pack(z.data_ptr(), 2, z.data_ptr(), 2, z.data_ptr(),
                                             z.data_ptr(), 1, 3, 2, z.data_ptr(), 0))


@pytest.mark.parametrize("world,K,nbins,tbins", [(3, 5, 4, 7), (8, 1000, 33, 65), (2, 1, 2, 2)])
def test_shard_message_kernels_32_bit_words(world, K, nbins, tbins):
    """dme_shard_pack32 / dme_shard_unpack32: every rank's message summed as int32 words (what the
    all-reduce does), then unpacked: bins widened exactly, status words summed (maxima taken),
    accumulators added in rank order bit for bit."""
    from bayespy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(world * 77 + K)
    nh, nv = K * nbins, tbins
    nmsg = nh + nv + 2 * world * (8 + K)
    assert L.dme_shard_message_words32(nh, nv, K, world) == nmsg
    stream = torch.cuda.current_stream().cuda_stream
    total = torch.zeros(nmsg, dtype=torch.int32, device="cuda")
    hsum, vsum = np.zeros(nh, dtype=np.int64), np.zeros(nv, dtype=np.int64)
    accs, stats = [], []
    top = (1 << 32) // world  # the sums reach beyond 2^31: the words are unsigned
    for rank in range(world):
        hist = rng.integers(0, top, size=nh, dtype=np.int64)
        histV = rng.integers(0, top, size=nv, dtype=np.int64)
        # status words near 2^32 and beyond: the ranks' words add in 64 bits, whatever their size
        status = rng.integers(0, 1 << 34, size=8, dtype=np.int64)
        acc = rng.random(K) * 10.0 ** rng.integers(-8, 8, size=K)
        if rank == 0:
            acc[0] = -acc[0]  # sign bit: the high word's top bit
        hsum += hist
        vsum += histV
        accs.append(acc)
        stats.append(status)
        d = [torch.from_numpy(a).cuda() for a in (hist, histV, status, acc)]
        msg = torch.full((nmsg,), -1, dtype=torch.int32, device="cuda")
        _lib.check(L.dme_shard_pack32(d[0].data_ptr(), nh, d[1].data_ptr(), nv, d[2].data_ptr(),
                                      d[3].data_ptr(), K, rank, world, msg.data_ptr(), stream))
        total += msg  # int32 wrap-around = the all-reduce's arithmetic
    out = torch.full((nh + nv + 8,), -1, dtype=torch.int64, device="cuda")
    acc_out = torch.empty(K, dtype=torch.float64, device="cuda")
    _lib.check(L.dme_shard_unpack32(total.data_ptr(), nh, nv, K, world, out.data_ptr(),
                                    out.data_ptr() + 8 * nh, out.data_ptr() + 8 * (nh + nv),
                                    acc_out.data_ptr(), stream))
    got = out.cpu().numpy()
    assert np.array_equal(got[:nh], hsum) and np.array_equal(got[nh:nh + nv], vsum)
    st = got[nh + nv:]
    summed = np.sum(stats, axis=0)
    assert np.array_equal(st[:4], summed[:4]) and np.array_equal(st[6:], summed[6:])
    assert st[4] == max(s[4] for s in stats) and st[5] == max(s[5] for s in stats)
    ref = accs[0].copy()
    for r in range(1, world):
        ref = ref + accs[r]
    assert np.array_equal(acc_out.cpu().numpy(), ref)


@pytest.mark.parametrize("K,nbins,wider", [(1, 2, 0), (7, 5, 0), (300, 700, 0), (5000, 40, 0), (2500, 9, 6)])
def test_compact_histogram_kernels_match_the_host_code(K, nbins, wider):
    """dme_shard_extents / _offsets / _compact / _expand against the torch code that host tensors
    (the gloo tests) take; `wider`: the agreed extents reach beyond this rank's bins (another rank
    saw larger counts), the compact words there are zeros and the way back fills a wider hist."""
    from bayespy_amd import parallel
    g = torch.Generator().manual_seed(K + nbins)
    hist = torch.randint(0, 3, (K, nbins), generator=g, dtype=torch.int64)
    hist *= (torch.rand((K, nbins), generator=g) < 0.2)
    hist[:, 0] = 0
    hist[K // 2] = 0
    hist[0, nbins - 1] = 5
    ext_h = torch.zeros(K + 2, dtype=torch.int64)
    ext_d = torch.zeros(K + 2, dtype=torch.int64, device="cuda")
    parallel._column_extents(hist.view(-1), K, nbins, ext_h)
    parallel._column_extents(hist.view(-1).cuda(), K, nbins, ext_d)
    assert torch.equal(ext_d.cpu(), ext_h)
    assert int(ext_h[0]) == nbins - 1 and (K == 1 or int(ext_h[K // 2]) == 0)
    if wider:
        ext_h[::3] += wider
        ext_d[::3] += wider
    off_h = torch.zeros(K + 1, dtype=torch.int64)
    off_d = torch.zeros(K + 1, dtype=torch.int64, device="cuda")
    parallel._column_offsets(ext_h, K, off_h)
    parallel._column_offsets(ext_d, K, off_d)
    assert torch.equal(off_d.cpu(), off_h)
    total = int(off_h[K])
    c_h = parallel._to_compact(hist.view(-1), K, nbins, off_h, total)
    c_d = parallel._to_compact(hist.view(-1).cuda(), K, nbins, off_d, total)
    assert c_h.numel() == total and torch.equal(c_d.cpu(), c_h)
    nb = nbins + wider
    back_h = parallel._from_compact(c_h, K, nb, off_h)
    back_d = parallel._from_compact(c_d, K, nb, off_d)
    assert torch.equal(back_d.cpu(), back_h)
    assert torch.equal(back_h.view(K, nb)[:, :nbins], hist) and not back_h.view(K, nb)[:, nbins:].any()


def _nccl_one_rank(rank, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bayespy_amd import device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rng = np.random.default_rng(3)
    K, N = 300, 20000
    X = (rng.integers(1, 40, size=(N, K)) * (rng.random((N, K)) < 0.05)).astype(np.uint16)
    X[5, 7] = 900
    Xd = torch.from_numpy(X).cuda()
    res = {}
    mt = int(X.sum(axis=1).max())
    info = {}
    for name, skw, kw in (
            ("words64", {}, {}), ("words32", {}, {"same_extents": True, "rows_bound": N}),
            ("compact", {}, {"compact": True, "rows_bound": N, "info": info}),
            # counts of 64 and more (the 900 among them) as records of an overflow store: they travel
            # inside the compact form and come back as records
            ("overflow", {"dense_bins": 64}, {"same_extents": True, "rows_bound": N}),
            # the compact message sized from a promised bound: no host read in the call
            ("bounded", {}, {"compact": True, "same_extents": True, "rows_bound": N,
                             "compact_bound": 40000}),
            ("bounded_overflow", {"dense_bins": 64}, {"same_extents": True, "compact_bound": 40000})):
        pend = []
        for _ in range(3):  # enqueued back to back, as bench.py does: no host wait in between
            st = device.DeviceStats(K, 900, mt, **skw)
            st.add_dense(Xd, max_count=900, max_total=mt)
            st.allreduce(force_message=True, **kw)
            pend.append(st.estimate_async(50))
        outs = [p.result() for p in pend]
        U, V = outs[-1]["uv"].to_lists()
        res[name] = (U, V, outs[-1]["info"]["rows"], outs[-1]["alpha"], [o["n_iters"] for o in outs])
    res["compact_words"] = info["compact_words"]
    # a compact bound that the statistics break: check() says so on every rank
    small = device.DeviceStats(K, 900, mt)
    small.add_dense(Xd, max_count=900, max_total=mt)
    small.allreduce(force_message=True, compact=True, same_extents=True, compact_bound=100)
    try:
        small.check()
        res["compact_bound_error"] = ""
    except RuntimeError as ex:
        res["compact_bound_error"] = str(ex)
    ref = device.DeviceStats(K, 900, int(X.sum(axis=1).max()))
    ref.add_dense(Xd, max_count=900, max_total=int(X.sum(axis=1).max()))
    o = ref.estimate_async(50).result()
    U, V = o["uv"].to_lists()
    res["plain"] = (U, V, o["info"]["rows"], o["alpha"], [o["n_iters"]] * 3)
    # a row bound that the job breaks: check() says so (a bin of the 32-bit message may have wrapped)
    low = device.DeviceStats(K, 900, int(X.sum(axis=1).max()))
    low.add_dense(Xd, max_count=900, max_total=int(X.sum(axis=1).max()))
    low.allreduce(force_message=True, same_extents=True, rows_bound=N - 1)
    try:
        low.check()
        res["bound_error"] = ""
    except ValueError as ex:
        res["bound_error"] = str(ex)
    out.update(res)
    dist.destroy_process_group()


def test_rccl_code_path_with_one_rank():
    """The nccl (= RCCL) branch of the all-reduce on real hardware: a group of one rank, the message
    forced through it (pack kernel -> RCCL all-reduce on the stream, no host wait -> unpack kernel ->
    scan -> enqueued solve), three estimations in flight at a time, in all three message forms.  The
    results are those of the same statistics without a collective."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nccl_one_rank, args=(port, out), nprocs=1, join=True)
    plain = out["plain"]
    for name in ("words64", "words32", "compact", "overflow", "bounded", "bounded_overflow"):
        U, V, rows, alpha, iters = out[name]
        assert U == plain[0] and V == plain[1] and rows == plain[2], name
        assert np.array_equal(alpha, plain[3]) and iters == plain[4], name
    assert "rows_bound" in out["bound_error"]
    assert 100 < out["compact_words"] <= 40000
    assert "compact_bound" in out["compact_bound_error"]
