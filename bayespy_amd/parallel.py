"""Row-sharded multi-GPU estimation: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on MI355X; "gloo" in CPU tests).

The statistics are sums of per-row indicator terms (DirichletEstimation/
dirichletMultinomialEstimation.py:116-123), so histograms built from disjoint row shards add
exactly in integers.  Communication per estimation: one MAX all-reduce of the two histogram
extents, one integer SUM all-reduce of hist || histV || status, one MAX of the two observed
maxima and one fp64 SUM of the K mean accumulators.  The suffix scan runs after the reduce and
the Newton loop is replicated on every rank — no per-iteration communication.
"""
import torch


def shard_rows(N, rank, world):
    """Contiguous row block [lo, hi) of rank `rank` out of `world` for N rows."""
    base, rem = divmod(int(N), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def grow_hist(hist, K, nbins, new_nbins):
    """hist viewed K x nbins -> zero-padded K x new_nbins (flattened)."""
    if new_nbins == nbins:
        return hist
    new = torch.zeros(K, new_nbins, dtype=hist.dtype, device=hist.device)
    new[:, :nbins] = hist.view(K, nbins)
    return new.view(-1)


def grow_vec(v, new_len):
    if new_len == v.numel():
        return v
    new = torch.zeros(new_len, dtype=v.dtype, device=v.device)
    new[:v.numel()] = v
    return new


def allreduce_stats_tensors(hist, histV, status, init_acc, K, nbins, tbins, group=None):
    """All-reduces one rank's partial statistics in place of a per-rank copy.  Tensors may live
    on any device the process group supports.  Returns (hist, histV, status, init_acc, nbins,
    tbins) holding the global statistics on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return hist, histV, status, init_acc, nbins, tbins
    ext = torch.tensor([nbins, tbins], dtype=torch.int64, device=hist.device)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=group)
    nb, tb = (int(v) for v in ext.cpu())
    hist = grow_hist(hist, K, nbins, nb)
    histV = grow_vec(histV, tb)
    mx = status[4:6].clone()
    flat = torch.cat([hist, histV, status])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    init_acc = init_acc.clone()
    dist.all_reduce(init_acc, op=dist.ReduceOp.SUM, group=group)
    nh, nv = hist.numel(), histV.numel()
    hist = flat[:nh].clone()
    histV = flat[nh:nh + nv].clone()
    status = flat[nh + nv:].clone()
    status[4:6] = mx
    return hist, histV, status, init_acc, nb, tb
