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


def _worker(rank, world, port, X, K, out):
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
    st.allreduce()
    info = st.check()
    uv = st.finalize()
    U, V = uv.to_lists()
    sol = uv.solve(st.cli_init_alphas(), 50, 0.0)
    out[rank] = (U, V, info["rows"], sol["alpha"], sol["n_iters"])
    dist.destroy_process_group()


def test_two_ranks_allreduce_then_identical_solve():
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
    mp.spawn(_worker, args=(2, port, X, fx["K"], out), nprocs=2, join=True)
    ref = [s for s in fx["solves"] if s["variant"] == "DE" and s["init_name"] == "cli"][0]
    for r in range(2):
        U, V, rows, alpha, n_iters = out[r]
        assert U == fx["U"] and V == fx["V"] and rows == fx["N"]
        np.testing.assert_allclose(alpha, ref["final"], rtol=1e-9)
        assert n_iters == ref["n_iters"]
    assert np.array_equal(out[0][3], out[1][3])  # replicated solve: bit-identical on every rank
