"""N > 1 path on CPU: two gloo ranks each hold the value histograms of their row shard (built by
the oracle's closed form); allreduce_stats_tensors must give every rank the histograms of the whole
matrix, integer-exact, whatever the per-rank extents."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hist_of(X, K):
    """Value histograms + mean accumulator of a row block (numpy; what the stats kernel emits)."""
    X = X.astype(np.int64)
    nb = int(X.max()) + 1 if X.size else 1
    tot = X.sum(axis=1)
    tb = int(tot.max()) + 1 if X.size else 1
    hist = np.zeros((K, nb), dtype=np.int64)
    for k in range(K):
        hist[k] = np.bincount(X[:, k], minlength=nb)
    hist[:, 0] = 0
    histV = np.bincount(tot, minlength=tb).astype(np.int64)
    histV[0] = 0
    acc = (X / (1.0 + tot)[:, None]).sum(axis=0)
    status = np.zeros(8, dtype=np.int64)
    status[0], status[4], status[5] = X.shape[0], X.max(initial=0), tot.max(initial=0)
    return hist, histV, status, acc, nb, tb


def _worker(rank, world, port, X, K, out, same_extents=False, rows_bound=None, compact=None):
    sys.path.insert(0, ROOT)
    from bayespy_amd.parallel import allreduce_stats_tensors, shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_rows(X.shape[0], rank, world)
    hist, histV, status, acc, nb, tb = _hist_of(X[lo:hi], K)
    if same_extents:  # every rank sizes its histograms for the same stated bounds
        from bayespy_amd.parallel import grow_hist, grow_vec
        hist = grow_hist(torch.from_numpy(hist.reshape(-1)), K, nb, 64).numpy()
        histV = grow_vec(torch.from_numpy(histV), 512).numpy()
        nb, tb = 64, 512
    res = allreduce_stats_tensors(torch.from_numpy(hist.reshape(-1)), torch.from_numpy(histV),
                                  torch.from_numpy(status), torch.from_numpy(acc), K, nb, tb,
                                  same_extents=same_extents, rows_bound=rows_bound,
                                  compact=compact, info=(info := {}))
    h, v, st, a, nb2, tb2 = res
    out[rank] = (h.view(K, nb2).numpy().copy(), v.numpy().copy(), st.numpy().copy(),
                 a.numpy().copy(), nb2, tb2, info)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_of_shard_histograms_is_exact(world):
    rng = np.random.default_rng(11)
    K, N = 6, 501
    X = rng.integers(0, 9, size=(N, K)) * (rng.random((N, K)) < 0.5)
    X[N - 1, 2] = 40  # only the last shard sees the largest count: extents differ between ranks
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), X, K, out), nprocs=world, join=True)
    hist, histV, status, acc, nb, tb = _hist_of(X, K)
    for r in range(world):
        h, v, st, a, nb2, tb2 = out[r][:6]
        assert (nb2, tb2) == (nb, tb)
        assert np.array_equal(h, hist) and np.array_equal(v, histV)
        assert st[0] == N and st[4] == 40 and st[5] == status[5]
        np.testing.assert_allclose(a, acc, rtol=1e-13)
        assert np.array_equal(a, out[0][3])  # summed in rank order: the same bits on every rank
    # and the suffix scan of the reduced histograms is the reference's U/V
    sys.path.insert(0, ROOT)
    import oracle
    ref = oracle.append_rows(X.astype(np.int32))
    for k in range(K):
        U_k = np.cumsum(hist[k][::-1])[::-1][1:]
        assert U_k[:len(ref.U[k])].tolist() == ref.U[k].tolist() and not U_k[len(ref.U[k]):].any()
    V = np.cumsum(histV[::-1])[::-1][1:]
    assert V[:len(ref.V)].tolist() == ref.V.tolist()


def test_allreduce_with_agreed_extents_is_one_collective():
    """same_extents=True: no extent round; results as above on histograms padded to the bounds."""
    rng = np.random.default_rng(12)
    K, N, world = 5, 300, 2
    X = rng.integers(0, 9, size=(N, K)) * (rng.random((N, K)) < 0.5)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), X, K, out, True), nprocs=world, join=True)
    hist, histV, status, acc, nb, tb = _hist_of(X, K)
    for r in range(world):
        h, v, st, a, nb2, tb2 = out[r][:6]
        assert (nb2, tb2) == (64, 512)
        assert np.array_equal(h[:, :nb], hist) and not h[:, nb:].any()
        assert np.array_equal(v[:tb], histV) and not v[tb:].any()
        assert st[0] == N and st[4] == status[4] and st[5] == status[5]
        np.testing.assert_allclose(a, acc, rtol=1e-13)
        assert np.array_equal(a, out[0][3])


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_in_32_bit_words_is_exact(world):
    """rows_bound < 2^32: the message travels in 32-bit words; same results, including
    a bin count and an accumulator bit pattern with the top bit of a 32-bit word set."""
    rng = np.random.default_rng(13)
    K, N = 4, 40
    X = rng.integers(0, 6, size=(N, K))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), X, K, out, True, N), nprocs=world, join=True)
    hist, histV, status, acc, nb, tb = _hist_of(X, K)
    for r in range(world):
        h, v, st, a, nb2, tb2 = out[r][:6]
        assert np.array_equal(h[:, :nb], hist) and not h[:, nb:].any()
        assert np.array_equal(v[:tb], histV) and not v[tb:].any()
        assert st[0] == N and st[4] == status[4] and st[5] == status[5]
        np.testing.assert_allclose(a, acc, rtol=1e-13)
        assert np.array_equal(a, out[0][3])


def _unbalanced_worker(rank, world, port, rows, out):
    sys.path.insert(0, ROOT)
    from bayespy_amd.parallel import allreduce_stats_tensors
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, nb, tb = 3, 4, 5
    hist = torch.zeros(K * nb, dtype=torch.int64)
    hist[1] = rows[rank]  # one bin holds every row of the shard
    histV = torch.zeros(tb, dtype=torch.int64)
    histV[2] = rows[rank]
    status = torch.tensor([rows[rank], 0, 0, 0, 1 + rank, 2 + rank, 0, 0], dtype=torch.int64)
    acc = torch.full((K,), 0.25 * (rank + 1), dtype=torch.float64)
    res = allreduce_stats_tensors(hist, histV, status, acc, K, nb, tb, same_extents=True,
                                  rows_bound=(1 << 32) - 1)
    out[rank] = (res[0].numpy().copy(), res[1].numpy().copy(), res[2].numpy().copy(), res[3].numpy().copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("rows", [(3_000_000_000, 1_000_000_000), (2_500_000_000, 1_000_000_000, 700_000_000)])
def test_32_bit_message_with_unbalanced_shards(rows):
    """A rank that holds more than 2^32 / world rows while the job stays below 2^32: every
    rank's status words travel exactly in its own slot row and are added in 64 bits, so the row count
    (what bench.py and DeviceStats.check() compare with the promise) and the bins are exact."""
    world = len(rows)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_unbalanced_worker, args=(world, _free_port(), rows, out), nprocs=world, join=True)
    for r in range(world):
        h, v, st, a = out[r]
        assert int(st[0]) == sum(rows) and int(h[1]) == sum(rows) and int(v[2]) == sum(rows)
        assert int(st[4]) == world and int(st[5]) == world + 1
        assert np.array_equal(a, out[0][3])


@pytest.mark.parametrize("world,rows_bound", [(2, None), (3, 1000)])
def test_allreduce_in_compact_form_is_exact(world, rows_bound):
    """compact=True: only the bins 1 .. ext[k] of every column travel (ext agreed by a MAX
    all-reduce together with the two histogram extents); one column holds a count far above the
    others' and only the last shard sees it, one column is empty; the results are those of the
    dense message and the message is a fraction of its size."""
    rng = np.random.default_rng(14)
    K, N = 40, 600
    X = rng.integers(0, 4, size=(N, K)) * (rng.random((N, K)) < 0.3)
    X[:, 7] = 0
    X[N - 1, 3] = 900
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), X, K, out, False, rows_bound, True), nprocs=world,
             join=True)
    hist, histV, status, acc, nb, tb = _hist_of(X, K)
    dense_words = K * nb + tb
    for r in range(world):
        h, v, st, a, nb2, tb2, info = out[r]
        assert (nb2, tb2) == (nb, tb)
        assert np.array_equal(h, hist) and np.array_equal(v, histV)
        assert st[0] == N and st[4] == 900 and st[5] == status[5]
        np.testing.assert_allclose(a, acc, rtol=1e-13)
        assert np.array_equal(a, out[0][3])
        word = 4 if rows_bound else 8
        assert info["message_bytes"] < word * dense_words / 10
        assert info["extents_bytes"] == (K + 3) * 8 and info["compact_words"] > 0
