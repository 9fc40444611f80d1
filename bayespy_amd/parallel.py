"""Row-sharded multi-GPU estimation: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on MI355X; "gloo" in CPU tests).

The statistics are sums of per-row indicator terms (DirichletEstimation/
dirichletMultinomialEstimation.py:116-123), so histograms built from disjoint row shards add
exactly in integers.  Communication per estimation: ONE integer SUM all-reduce (histograms, status
words, and per-rank slots that transport the maxima and the fp64 mean accumulators unchanged),
preceded by one MAX all-reduce of the two histogram extents unless the ranks already agree on
them.  The suffix scan runs after the reduce and the Newton loop is replicated on every rank —
no per-iteration communication.
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


COMPACT_MIN_K = 4096  # from this many columns on the count histograms travel in compact form,
COMPACT_MIN_WORDS = 1 << 20  # or from this many bins on when the ranks agree on nbins beforehand
                             # (heavy-tailed counts on few columns: BASELINE configs[4], 100 x 1e4)


def allreduce_stats_tensors(hist, histV, status, init_acc, K, nbins, tbins, group=None,
                            same_extents=False, rows_bound=None, compact=None, info=None,
                            force_message=False, overflow=None, compact_bound=None):
    """All-reduces one rank's partial statistics.  Tensors may live on any device the process
    group supports.  Returns (hist, histV, status, init_acc, nbins, tbins) holding the global
    statistics on every rank.

    One integer SUM carries everything (layout of include/dme_hip.h, dme_shard_pack):
    hist || histV || status[8] (the two maxima, [4:6], as 0) || one slot row per rank for its two
    observed maxima and the bit patterns of its K mean accumulators.  A rank fills only its own
    slot row, so the sum hands every rank all the per-rank values unchanged; the maxima are then
    taken locally and the fp64 accumulators are added in rank order, which makes the start point of
    the replicated solve identical on every rank and from run to run.  On the GPU the message is
    written and read by two HIP kernels (dme_shard_pack / dme_shard_unpack); host tensors (the
    gloo tests) build the same message with torch.  `same_extents` (every rank built its histograms
    with the same max_count / max_total) skips the MAX all-reduce that otherwise agrees the
    extents first.

    `rows_bound`: an upper bound on the rows of the whole job that EVERY rank passes alike (ranks that
    pass different bounds build messages of different sizes: the collective then fails or hangs).
    Below 2^32 no bin can need more than 32 bits, and the message travels in 32-bit words (bins one
    word each; every rank's status words and accumulators in its own slot row as (lo, hi) pairs, summed
    in 64 bits afterwards: dme_shard_pack32 / dme_shard_unpack32): half the bytes on the wire (SURVEY
    8e "uint32 if N < 2^32").  The promise is checked when the statistics are: DeviceStats.check()
    raises if the job counted more rows than the bound (a bin may then have wrapped).

    `compact` (default: K >= COMPACT_MIN_K, or same_extents and K * nbins >= COMPACT_MIN_WORDS; the
    same on every rank): the count histograms travel as
    the bins 1 .. ext[k] of every column laid end to end, ext[k] the column's largest occupied bin
    over all ranks (one MAX all-reduce of K + 2 words, which also agrees the extents, and one host
    read of the compact length).  With many columns nearly all of them stop far below the largest
    count of any: BASELINE configs[3] (K = 50 000) sends a few MB instead of K x nbins words.

    `overflow`: the DeviceStats that owns the histograms (its `.overflow` store holds the counts
    beyond the dense bins, include/dme_hip.h).  A store on any rank makes the message compact: the
    records are added into the compact form before the sum and the merged cells beyond the dense
    bins are written back as records afterwards.
    `compact_bound` (GPU tensors, with same_extents; the same on every rank): a promised upper bound
    on the compact length.  The message is sized from it and the lengths stay on the device - no host
    read in the whole call; a compact form that outgrows the bound raises status[7], which travels
    in the message and is reported by DeviceStats.check() on every rank.

    `info` (a dict, optional) receives message_bytes and, in compact form, extents_bytes and (measured
    path) compact_words.
    `force_message`: build, reduce and read the message even in a group of one rank (rehearsals of
    the RCCL code path on a single GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return hist, histV, status, init_acc, nbins, tbins
    if dist.get_world_size(group) == 1 and not force_message:
        return hist, histV, status, init_acc, nbins, tbins
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    store = overflow.overflow if overflow is not None else None
    if compact is None:
        # (a rule every rank evaluates alike: K always is the same, nbins only when agreed beforehand)
        compact = K >= COMPACT_MIN_K or (same_extents and K * nbins >= COMPACT_MIN_WORDS)
    if same_extents and store is not None:
        compact = True  # (the same bounds on every rank: every rank has a store)
    nb, tb = nbins, tbins
    if not compact and not same_extents:
        # the histogram extents - and whether any rank keeps an overflow store, which makes the
        # message compact on every rank - agreed in one MAX
        ext = torch.tensor([nbins, tbins, 0 if store is None else 1], dtype=torch.int64,
                           device=hist.device)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=group)
        nb, tb, any_store = (int(v) for v in ext.cpu())
        if any_store:
            compact = True
        else:
            hist = grow_hist(hist, K, nbins, nb)
            histV = grow_vec(histV, tb)
    off = None
    if compact:
        # the columns' extents, and with them the two histogram extents, agreed in one MAX
        ext = torch.empty(K + 3, dtype=torch.int64, device=hist.device)
        ext[K], ext[K + 1], ext[K + 2] = nbins, tbins, 0 if store is None else 1
        _column_extents(hist, K, nbins, ext, store)
        _sync_for_gloo(ext, group)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=group)
        off = torch.empty(K + 1, dtype=torch.int64, device=hist.device)
        _column_offsets(ext, K, off)
        if compact_bound is not None and same_extents and hist.is_cuda:
            total = int(compact_bound)  # promised: nothing is read back
        else:
            nb, tb, any_store, total = (int(v) for v in
                                        torch.stack([ext[K], ext[K + 1], ext[K + 2], off[K]]).cpu())
            if any_store and store is None and overflow is not None:
                overflow._need_overflow()  # another rank's records will come back to this one
                store = overflow.overflow
            if info is not None:
                info["compact_words"] = total
        histV = grow_vec(histV, tb)
        hist = _to_compact(hist, K, nbins, off, total, store, status)
        if info is not None:
            info["extents_bytes"] = (K + 3) * 8
    out = _sum_message(hist, histV, status, init_acc, K, group, world, rank, rows_bound, info)
    if compact:
        out = (_from_compact(out[0], K, nb, off, store, out[2]),) + out[1:]
    return out + (nb, tb)


def _sync_for_gloo(t, group):
    """gloo only (CPU tests never reach it; rehearsals of bench.py with gloo ranks on a GPU do): drain
    the current stream before handing gloo a device tensor.

    What is known (round 3, reproduced on an MI355X box with DME_GLOO_NOSYNC=1 under bench.py's
    watchdog; profiles/r3_notes.md):
      * without this wait the process blocks for good inside dist.all_reduce -> ProcessGroupGloo's
        work.wait(), also with ONE rank (so it is not two processes sharing a card), and only when
        steps are enqueued back to back: the step whose all-reduce hangs finds the previous step's
        solver launches and its 128-byte copy to pinned memory still queued on the current stream.
        A step that starts on a drained stream (warm-up, `--sync-steps`) goes through, and so do small
        inputs, whose kernels finish before the host arrives (100 000 rows: fine; 4 M rows: hangs);
      * gloo does not run on our stream: ProcessGroupGloo records an event on the current stream
        (behind the pack kernel), makes a stream of ITS OWN from torch's high-priority pool wait for
        it, copies the tensor to pinned host memory there, and a worker thread blocks in that stream's
        synchronize before the CPU reduction (ProcessGroupGloo.cpp: initializeStreamsEvents /
        AsyncAllreduceCUDAWork).  The wait that never returns is that side stream's: a cross-stream
        event wait issued while the recording stream is still busy.  Why the ROCm runtime does not
        resolve it here is not established; the pack kernel, the message and the collective's
        arguments are the same ones the RCCL path uses;
      * RCCL ("nccl") needs no host wait and has none: ProcessGroupNCCL enqueues the collective on its
        own stream behind the same kind of event and no host thread waits on it -
        tests/test_gpu_distributed.py::test_rccl_code_path_with_one_rank runs three estimations in
        flight through it, in every message form."""
    import os
    import torch.distributed as dist
    if os.environ.get("DME_GLOO_NOSYNC"):  # (the diagnosis above)
        return
    if t.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(t.device).synchronize()


def _column_extents(hist, K, nbins, ext, store=None):
    """ext[k] = the largest occupied bin of column k, overflow records included (dme_shard_extents,
    dme_overflow_extents; torch for host tensors)."""
    if hist.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        with _on(hist.device):
            _lib.check(_lib.lib().dme_shard_extents(_ptr(hist), K, nbins, _ptr(ext), _stream()))
            if store is not None:
                _lib.check(_lib.lib().dme_overflow_extents(_ptr(store), K, _ptr(ext), _stream()))
    else:
        bins = torch.arange(nbins, dtype=torch.int64)
        ext[:K] = ((hist.view(K, nbins) != 0) * bins).max(dim=1).values


def _column_offsets(ext, K, off):
    if ext.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        with _on(ext.device):
            _lib.check(_lib.lib().dme_shard_offsets(_ptr(ext), K, _ptr(off), _stream()))
    else:
        off[0] = 0
        off[1:] = torch.cumsum(ext[:K], 0)


def _to_compact(hist, K, nbins, off, total, store=None, status=None):
    """`total` words of compact form: the measured length, or a promised bound (GPU tensors; the words
    behind the true length are zeroed, a longer form raises status[7])."""
    out = torch.empty(max(total, 1), dtype=torch.int64, device=hist.device)[:total]
    if hist.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        if total:
            with _on(hist.device):
                _lib.check(_lib.lib().dme_shard_compact(_ptr(hist), K, nbins, _ptr(store), _ptr(off),
                                                        _ptr(out), total, _ptr(status), _stream()))
        return out
    if total == 0:
        return out
    n = (off[1:] - off[:-1])
    wide = max(nbins, int(n.max()) + 1)
    h = grow_hist(hist, K, nbins, wide).view(K, wide)
    bins = torch.arange(wide, dtype=torch.int64)
    return h[(bins[None, :] >= 1) & (bins[None, :] <= n[:, None])].contiguous()


def _from_compact(comp, K, nbins, off, store=None, status=None):
    hist = torch.empty(K * nbins, dtype=torch.int64, device=comp.device)
    if comp.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        if comp.numel() == 0:
            return hist.zero_()
        L = _lib.lib()
        with _on(comp.device):
            if store is not None:
                # the records went into the compact form; the merged cells beyond the dense bins come
                # back as fresh records
                _lib.check(L.dme_overflow_init(_ptr(store), (store.numel() - 2) // 2, _stream()))
            _lib.check(L.dme_shard_expand(_ptr(comp), comp.numel(), K, nbins, _ptr(off), _ptr(hist),
                                          _ptr(store), _ptr(status), _stream()))
        return hist
    hist.zero_()
    n = (off[1:] - off[:-1])
    bins = torch.arange(nbins, dtype=torch.int64)
    hist.view(K, nbins)[(bins[None, :] >= 1) & (bins[None, :] <= n[:, None])] = comp
    return hist


def _sum_message(hist, histV, status, init_acc, K, group, world, rank, rows_bound, info):
    """The SUM all-reduce of one message; `hist` is any flat array of bins (K x nbins, or compact)."""
    import torch.distributed as dist
    nh, nv = hist.numel(), histV.numel()
    if rows_bound is not None and 0 <= int(rows_bound) < (1 << 32):
        if info is not None:
            info["message_bytes"] = 4 * (nh + nv + 2 * world * (8 + K))
        return _allreduce_words32(hist, histV, status, init_acc, K, group, world, rank)
    nmsg = nh + nv + 8 + world * (2 + K)
    if info is not None:
        info["message_bytes"] = 8 * nmsg
    if hist.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        L = _lib.lib()
        with _on(hist.device):
            msg = torch.empty(nmsg, dtype=torch.int64, device=hist.device)
            _lib.check(L.dme_shard_pack(_ptr(hist), nh, _ptr(histV), nv, _ptr(status),
                                        _ptr(init_acc), K, rank, world, _ptr(msg), _stream()))
            _sync_for_gloo(msg, group)
            dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)
            acc = torch.empty(K, dtype=torch.float64, device=hist.device)
            _lib.check(L.dme_shard_unpack(_ptr(msg), nh, nv, K, world, _ptr(acc), _stream()))
        return msg[:nh], msg[nh:nh + nv], msg[nh + nv:nh + nv + 8], acc
    msg = torch.zeros(nmsg, dtype=torch.int64)
    msg[:nh] = hist
    msg[nh:nh + nv] = histV
    msg[nh + nv:nh + nv + 8] = status
    msg[nh + nv + 4:nh + nv + 6] = 0  # the maxima travel in the slots
    slots = msg[nh + nv + 8:].view(world, 2 + K)
    slots[rank, :2] = status[4:6]
    slots[rank, 2:] = init_acc.view(torch.int64)
    dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)
    out_status = msg[nh + nv:nh + nv + 8]
    out_status[4:6] = slots[:, :2].max(dim=0).values
    parts = slots[:, 2:].contiguous().view(torch.float64)
    acc = parts[0].clone()
    for r in range(1, world):  # fixed order: bit-identical on every rank
        acc += parts[r]
    return msg[:nh], msg[nh:nh + nv], out_status, acc


def _allreduce_words32(hist, histV, status, init_acc, K, group, world, rank):
    """The 32-bit form of the message (include/dme_hip.h, dme_shard_pack32): int32 SUM all-reduce."""
    import torch.distributed as dist
    nh, nv = hist.numel(), histV.numel()
    nmsg = nh + nv + 2 * world * (8 + K)
    if hist.is_cuda:
        from . import _lib
        from .device import _on, _ptr, _stream
        L = _lib.lib()
        with _on(hist.device):
            msg = torch.empty(nmsg, dtype=torch.int32, device=hist.device)
            _lib.check(L.dme_shard_pack32(_ptr(hist), nh, _ptr(histV), nv, _ptr(status),
                                          _ptr(init_acc), K, rank, world, _ptr(msg), _stream()))
            _sync_for_gloo(msg, group)
            dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)
            out = torch.empty(nh + nv + 8, dtype=torch.int64, device=hist.device)
            acc = torch.empty(K, dtype=torch.float64, device=hist.device)
            _lib.check(L.dme_shard_unpack32(_ptr(msg), nh, nv, K, world, _ptr(out),
                                            out.data_ptr() + 8 * nh, out.data_ptr() + 8 * (nh + nv),
                                            _ptr(acc), _stream()))
        return out[:nh], out[nh:nh + nv], out[nh + nv:], acc
    # host tensors (gloo tests): the same message with torch
    tail = torch.zeros(world * (8 + K), dtype=torch.int64)
    slots = tail.view(world, 8 + K)
    slots[rank, :8] = status
    slots[rank, 8:] = init_acc.view(torch.int64)
    lohi = torch.stack([tail & 0xFFFFFFFF, (tail >> 32) & 0xFFFFFFFF], dim=1).reshape(-1)
    msg = torch.cat([hist, histV, lohi])  # int64 holding 32-bit words
    msg = (msg & 0xFFFFFFFF).to(torch.int64)
    msg32 = torch.where(msg >= (1 << 31), msg - (1 << 32), msg).to(torch.int32)
    dist.all_reduce(msg32, op=dist.ReduceOp.SUM, group=group)
    words = msg32.to(torch.int64) & 0xFFFFFFFF
    t = words[nh + nv:].view(-1, 2)
    tail = t[:, 0] | (t[:, 1] << 32)
    slots = tail.view(world, 8 + K)
    out_status = slots[:, :8].sum(dim=0)  # 64-bit sums of the ranks' own words: exact
    out_status[4:6] = slots[:, 4:6].max(dim=0).values
    parts = slots[:, 8:].contiguous().view(torch.float64)
    acc = parts[0].clone()
    for r in range(1, world):  # fixed order: bit-identical on every rank
        acc += parts[r]
    return words[:nh].clone(), words[nh:nh + nv].clone(), out_status, acc
