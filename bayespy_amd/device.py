"""Device-resident statistics and solver front-end (host side of the C ABI in include/dme_hip.h).

torch is used for device buffers, streams and torch.distributed only; every computation on the
hot path is a HIP kernel in libdme_hip.so.  There is no CPU fallback: without a GPU these classes
raise.

Reference mapping (DirichletEstimation/dirichletMultinomialEstimation.py = "DE/dme.py"):
    DeviceStats.add_dense / add_csr   <- CompressedRowData.appendRow over all rows (DE/dme.py:113-123)
                                         + the CLI mean accumulator (DE/fdp.py:87-89)
    DeviceStats.finalize              <- the U/V lists those calls leave behind
    DeviceUV.evaluate                 <- dirichLogProb / priorGradient / priorHessian* (:16-61)
    DeviceUV.solve                    <- findDirichletPriors (:126-190)
"""
import contextlib
import ctypes
import logging

import numpy as np
import torch

from . import _lib
from ._trace import phase as _phase

TRACE_CAP = 4096  # iterations whose trace (loss, gradient size, branch, alphas) is kept

# The count histograms are dense K x nbins only up to DENSE_CELLS cells (256 MB of 64-bit bins): nbins
# stops at max(DENSE_MIN_BINS, DENSE_CELLS // K) and larger counts are records of an overflow store
# (include/dme_hip.h).  From RAGGED_MIN_CELLS cells on - or whenever a store is in use - finalize()
# lays U/V out ragged, every column at its own length, as the reference's lists are (DE/dme.py:118).
DENSE_CELLS = 1 << 25
DENSE_MIN_BINS = 64
RAGGED_MIN_CELLS = 1 << 22
OVERFLOW_CAPACITY = 1 << 16  # records of a first overflow store (1 MB); see DeviceStats.overflow_capacity

EXIT_MESSAGES = {
    1: "Converged with small gradient",    # DE/dme.py:146
    2: "Converged with small learn rate",  # DE/dme.py:183
    3: "Reached max iterations",           # DE/dme.py:189
}


_GPU_CHECKED = False


def require_gpu(device=None):
    global _GPU_CHECKED
    if not _GPU_CHECKED:  # once: the answer does not change within a process
        if not torch.cuda.is_available():
            raise RuntimeError("bayespy_amd needs an AMD GPU (MI355X, ROCm): no HIP device is "
                               "visible and there is no CPU fallback")
        _lib.lib()
        _GPU_CHECKED = True
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("bayespy_amd runs on the GPU only")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


_SAME_DEVICE = contextlib.nullcontext()


def _on(device):
    """Context that makes `device` current; a no-op (no driver calls) when it already is."""
    return _SAME_DEVICE if torch.cuda.current_device() == device.index else torch.cuda.device(device)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current stream of the current device as a raw hipStream_t (called before every launch:
    the direct binding costs a fraction of a microsecond, torch.cuda.current_stream() several)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class DeviceUV:
    """The reference's compressed statistics on the device — what every solver entry point consumes.

    Two layouts of the fp64 U (and of its exact int64 twin `Ui`):
      dense-padded  K x ldU, U[k][n] at k * ldU + n (zero padding is neutral in every sum);
      ragged        every column at its own length, U[k][n] at off[k] + n (`ragged` holds the offset and
                    slot tables of include/dme_hip.h, dme_ragged_t) — the reference's lists are ragged
                    too (DE/dme.py:118), and one outlier count then costs one long column, not K of
                    them.
    lenU[K], V, lenV are the same in both."""

    def __init__(self, K, Uf, ldU, lenU, Vf, lenV, U=None, V=None, ragged=None):
        self.K = int(K)
        self.Uf, self.ldU, self.lenU = Uf, int(ldU), lenU
        self.Vf, self.lenV = Vf, lenV
        self.capV = int(Vf.numel())
        self.Ui, self.Vi = U, V  # exact int64 copies when built by the scan
        self.device = Uf.device
        self._ws = None
        # ragged layout: (off[K+1], slotoff[K+1], slotcol[nslots], nslots, total) - device tensors and
        # two host integers; self._rg is the C struct handed to the *_ragged entries
        self.ragged = ragged
        self._rg = None
        if ragged is not None:
            off, slotoff, slotcol, nslots, total = ragged
            self._rg = _lib.Ragged(off.data_ptr(), slotoff.data_ptr(), slotcol.data_ptr(), int(nslots),
                                   int(total))
            self.nslots = int(nslots)

    @property
    def shape_key(self):
        """What dme_solve_finish files the launch-count hint under (ldU, or the ragged slot count)."""
        return self.nslots if self.ragged is not None else self.ldU

    def _rgp(self):
        return ctypes.addressof(self._rg)

    # -- construction from host-side ragged lists (foreign data objects with .U/.V) ------------
    @classmethod
    def from_lists(cls, U, V, device=None):
        device = require_gpu(device)
        K = len(U)
        ldU = max(1, max((len(u) for u in U), default=0))
        capV = max(1, len(V))
        vhost = np.zeros(capV, dtype=np.float64)
        if len(V):
            vhost[:len(V)] = np.asarray(V, dtype=np.float64)
        lenU = np.array([len(u) for u in U], dtype=np.int64)
        lenV = torch.tensor([len(V)], dtype=torch.int64, device=device)
        if K * ldU >= RAGGED_MIN_CELLS and int(lenU.sum()) * 2 < K * ldU:
            # long lists of very different lengths: kept ragged, as they arrive
            seg = _lib.SEGMENT_TERMS
            off = np.zeros(K + 1, dtype=np.int64)
            np.cumsum(lenU, out=off[1:])
            cnt = (lenU + seg - 1) // seg
            slotoff = np.zeros(K + 1, dtype=np.int64)
            np.cumsum(cnt, out=slotoff[1:])
            total, nslots = int(off[K]), int(slotoff[K])
            host = np.zeros(max(total, 1), dtype=np.float64)
            for k, u in enumerate(U):
                if len(u):
                    host[off[k]:off[k + 1]] = np.asarray(u, dtype=np.float64)
            slotcol = np.repeat(np.arange(K, dtype=np.int32), cnt)
            if nslots == 0:
                slotcol = np.zeros(1, dtype=np.int32)
            rg = (torch.from_numpy(off).to(device), torch.from_numpy(slotoff).to(device),
                  torch.from_numpy(slotcol).to(device), nslots, total)
            return cls(K, torch.from_numpy(host).to(device), 1, torch.from_numpy(lenU).to(device),
                       torch.from_numpy(vhost).to(device), lenV, ragged=rg)
        host = np.zeros((K, ldU), dtype=np.float64)
        for k, u in enumerate(U):
            if len(u):
                host[k, :len(u)] = np.asarray(u, dtype=np.float64)
        return cls(K, torch.from_numpy(host).to(device), ldU, torch.from_numpy(lenU).to(device),
                   torch.from_numpy(vhost).to(device), lenV)

    def workspace(self):
        if self._ws is None:
            if self.ragged is not None:
                nbytes = _lib.lib().dme_solver_workspace_bytes_ragged(self.K, self.nslots, self.capV)
            else:
                nbytes = _lib.lib().dme_solver_workspace_bytes(self.K, self.ldU, self.capV)
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off, self._ws.numel() - off

    def to_lists(self):
        """(U, V) as the reference holds them: ragged lists of ints."""
        lenU = self.lenU.cpu().numpy()
        lenV = int(self.lenV.cpu()[0])
        exact = self.Ui is not None
        Uh = (self.Ui if exact else self.Uf).cpu().numpy()
        Vh = (self.Vi if exact else self.Vf).cpu().numpy()
        if self.ragged is not None:
            off = self.ragged[0].cpu().numpy()
            rows = [Uh[off[k]:off[k] + lenU[k]] for k in range(self.K)]
        else:
            Uh = Uh.reshape(self.K, self.ldU)
            rows = [Uh[k, :lenU[k]] for k in range(self.K)]
        if exact:
            return [[int(x) for x in r] for r in rows], [int(x) for x in Vh[:lenV]]
        return [r.tolist() for r in rows], Vh[:lenV].tolist()

    def dense_U(self):
        """The exact U as a dense K x max(len) int64 tensor on the device, whatever the layout (tests
        and checks; a torch scatter, off the hot path)."""
        if self.Ui is None:
            raise ValueError("no exact integer U (weighted statistics)")
        if self.ragged is None:
            return self.Ui.view(self.K, self.ldU)
        off = self.ragged[0]
        ext = off[1:] - off[:-1]
        width = max(int(ext.max()), 1)
        out = torch.zeros(self.K, width, dtype=torch.int64, device=self.device)
        total = int(off[self.K])
        if total:
            col = torch.repeat_interleave(torch.arange(self.K, device=self.device), ext)
            pos = torch.arange(total, device=self.device) - off[:-1][col]
            out[col, pos] = self.Ui[:total]
        return out

    # -- one evaluation: loss, gradient, Hessian parts (DE/dme.py:16-61) -------------------------
    def _beta(self, Beta):
        if Beta is None:
            return None
        b = torch.as_tensor(np.asarray(Beta, dtype=np.float64)).to(self.device)
        if b.numel() < self.K:
            raise IndexError("list index out of range")  # Beta[i] at CPT/dme.py:34
        return b[:self.K].contiguous()

    def evaluate(self, alpha, hyperprior=0.0, Beta=None, W=None):
        """Returns (logprob, hconst, grad[K], hdiag[K], gradSq) as host values.  With Beta / W the
        (Beta, W) conjugate hyperprior of CPT/dme.py:20-87 replaces the scalar one."""
        with _on(self.device):
            a = torch.as_tensor(np.asarray(alpha, dtype=np.float64)).to(self.device)
            if a.numel() != self.K:
                raise ValueError(f"expected {self.K} alphas, got {a.numel()}")
            scal = torch.empty(4, dtype=torch.float64, device=self.device)
            grad = torch.empty(self.K, dtype=torch.float64, device=self.device)
            hdiag = torch.empty(self.K, dtype=torch.float64, device=self.device)
            ws, wsb = self.workspace()
            cpt = not (Beta is None and W is None)
            if cpt and hyperprior:
                raise ValueError("the scalar hyperprior and (Beta, W) do not combine")
            b = self._beta(Beta) if cpt else None
            w, has_w = (0.0 if W is None else float(W)), (0 if W is None else 1)
            if self.ragged is not None:
                _lib.check(_lib.lib().dme_eval_ragged(
                    _ptr(self.Uf), self._rgp(), _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), float(hyperprior), _ptr(b), w, has_w, _ptr(scal),
                    _ptr(grad), _ptr(hdiag), ws, wsb, _stream()))
            elif not cpt:
                _lib.check(_lib.lib().dme_eval(
                    _ptr(self.Uf), self.ldU, _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), float(hyperprior), _ptr(scal), _ptr(grad),
                    _ptr(hdiag), ws, wsb, _stream()))
            else:
                _lib.check(_lib.lib().dme_eval_cpt(
                    _ptr(self.Uf), self.ldU, _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), _ptr(b), w, has_w, _ptr(scal), _ptr(grad),
                    _ptr(hdiag), ws, wsb, _stream()))
            s = scal.cpu().numpy()
            return float(s[0]), float(s[1]), grad.cpu().numpy(), hdiag.cpu().numpy(), float(s[3])

    # -- the Newton loop (DE/dme.py:126-190) ----------------------------------------------------
    def solve(self, init_alphas, iterations, hyperprior=0.0, trace_alphas=True, Beta=None, W=None,
              consume_init=False):
        """Runs findDirichletPriors on the device.  Returns a dict: alpha (np.float64[K]),
        exit (1/2/3), n_iters, n_evals, final_loss, trace (n_iters x 4: loss, gradSq, branch,
        learnRate), trace_alpha (n_iters x K or None).  With Beta / W: CPT/dme.py:156-219.
        `consume_init`: a float64 device tensor passed as init_alphas is overwritten with the result
        instead of being copied first (the caller gives it up; saves a device copy)."""
        iterations = int(iterations)
        with _on(self.device), _phase("solve"):
            if (consume_init and isinstance(init_alphas, torch.Tensor)
                    and init_alphas.device == self.device and init_alphas.dtype == torch.float64
                    and init_alphas.is_contiguous()):
                a = init_alphas.detach()
            elif isinstance(init_alphas, torch.Tensor):
                a = init_alphas.detach().to(device=self.device, dtype=torch.float64).clone()
            else:
                a = torch.as_tensor(np.array(init_alphas, dtype=np.float64)).to(self.device)
            if a.numel() != self.K:
                raise ValueError(f"expected {self.K} alphas, got {a.numel()}")
            # the trace holds the first TRACE_CAP iterations (the reference's cost depends on the
            # iterations actually run, typically ~10; `-i 1000000` must not allocate gigabytes)
            cap = min(max(iterations, 1), TRACE_CAP)
            # one buffer: result[8] int64 | final loss | trace[cap][4]   (+ optional alphas trace)
            buf = torch.empty(8 + 1 + cap * 4, dtype=torch.float64, device=self.device)
            result = buf[:8].view(torch.int64)
            floss = buf[8:9]
            trace = buf[9:]
            talpha = (torch.empty(cap * self.K, dtype=torch.float64, device=self.device)
                      if trace_alphas else None)
            host = (ctypes.c_int64 * 8)()
            ws, wsb = self.workspace()
            cpt = not (Beta is None and W is None)
            if cpt and hyperprior:
                raise ValueError("the scalar hyperprior and (Beta, W) do not combine")
            b = self._beta(Beta) if cpt else None
            w, has_w = (0.0 if W is None else float(W)), (0 if W is None else 1)
            if self.ragged is not None:
                _lib.check(_lib.lib().dme_solve_ragged(
                    _ptr(self.Uf), self._rgp(), _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), iterations, float(hyperprior), _ptr(b), w, has_w,
                    _ptr(trace), _ptr(talpha), cap, _ptr(result), ctypes.addressof(host), _ptr(floss),
                    ws, wsb, _stream()))
            elif not cpt:
                _lib.check(_lib.lib().dme_solve(
                    _ptr(self.Uf), self.ldU, _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), iterations, float(hyperprior), _ptr(trace),
                    _ptr(talpha), cap, _ptr(result), ctypes.addressof(host), _ptr(floss), ws, wsb,
                    _stream()))
            else:
                _lib.check(_lib.lib().dme_solve_cpt(
                    _ptr(self.Uf), self.ldU, _ptr(self.lenU), _ptr(self.Vf), self.capV,
                    _ptr(self.lenV), self.K, _ptr(a), iterations, _ptr(b), w, has_w, _ptr(trace),
                    _ptr(talpha), cap, _ptr(result), ctypes.addressof(host), _ptr(floss), ws, wsb,
                    _stream()))
            # dme_solve synchronised the stream and filled `host`: no further device round trip
            return SolveResult(a, list(host), trace.view(cap, 4), talpha, cap, self.K)


class SolveResult(dict):
    """What findDirichletPriors returned, plus the trace.  Scalars are host values already;
    arrays stay on the device until asked for (`alpha`, `trace`, `trace_alpha` copy on first use)."""

    def __init__(self, alpha_dev, res, trace_dev, talpha_dev, cap, K):
        import struct
        super().__init__()
        self._lazy = {
            "alpha": lambda: alpha_dev.cpu().numpy(),
            "trace": lambda: trace_dev[:min(res[1], cap)].cpu().numpy(),
            "trace_alpha": (lambda: talpha_dev.view(cap, K)[:min(res[1], cap)].cpu().numpy())
            if talpha_dev is not None else (lambda: None),
        }
        dict.update(self, {
            "alpha_device": alpha_dev,
            "exit": int(res[0]),
            "n_iters": int(res[1]),
            "n_evals": int(res[2]),
            "launches": int(res[4]),
            "final_loss": struct.unpack("d", struct.pack("q", res[5]))[0],
            # accept / reject decisions taken on a loss difference of <= ~4 ulp
            "fragile_decisions": int(res[7]),
        })

    def __missing__(self, key):
        if key in self._lazy:
            val = self._lazy[key]()
            dict.__setitem__(self, key, val)
            return val
        raise KeyError(key)


class DeviceStats:
    """Value histograms of a count matrix, accumulated on the device.

    hist[k][v] = #rows with x_ik == v, histV[t] = #rows with total t, init_acc[k] = the CLI's
    mean accumulator.  Row chunks and row shards (GPUs) simply add, so `allreduce()` is one integer
    sum; `finalize()` runs the reverse suffix scan that yields the reference's U/V."""

    def __init__(self, K, max_count=1, max_total=1, device=None, dense_bins=None,
                 overflow_capacity=OVERFLOW_CAPACITY):
        """dense_bins: the widest dense count histogram (bins per column, bin 0 included; default
        max(64, 2^25 // K)); counts beyond it are records of the overflow store, which holds
        `overflow_capacity` of them (a full store reports the cells it had to drop: check() raises)."""
        self.device = require_gpu(device)
        self.K = int(K)
        self.dense_bins = (int(dense_bins) if dense_bins is not None
                           else max(DENSE_MIN_BINS, DENSE_CELLS // max(self.K, 1)))
        if self.dense_bins < 2:
            raise ValueError("dense_bins must be at least 2")
        self.overflow_capacity = int(overflow_capacity)
        self.overflow = None  # int64 tensor [2 + 2 * capacity]: head + records (dme_overflow_*)
        self.nbins = min(max(int(max_count), 1) + 1, self.dense_bins)
        self.tbins = max(int(max_total), 1) + 1
        with _on(self.device):
            # one zero-filled buffer (one memset): hist | histV | status | init_acc (fp64 zeros are
            # all-zero bits)
            nh, nv = self.K * self.nbins, self.tbins
            buf = torch.zeros(nh + nv + 8 + self.K, dtype=torch.int64, device=self.device)
            self.hist = buf[:nh]
            self.histV = buf[nh:nh + nv]
            self.status = buf[nh + nv:nh + nv + 8]
            self.init_acc = buf[nh + nv + 8:].view(torch.float64)
        self._ws = None
        self._csr_ws = None
        if int(max_count) + 1 > self.dense_bins:
            self._need_overflow()

    def _need_overflow(self):
        if self.overflow is None:
            L = _lib.lib()
            cap = max(self.overflow_capacity, 1)
            with _on(self.device):
                self.overflow = torch.empty(L.dme_overflow_bytes(cap) // 8, dtype=torch.int64,
                                            device=self.device)
                _lib.check(L.dme_overflow_init(_ptr(self.overflow), cap, _stream()))

    # -- capacity -------------------------------------------------------------------------------
    def reserve(self, max_count, max_total):
        """Grow the histograms so that counts <= max_count and totals <= max_total fit: the dense
        bins up to `dense_bins`, an overflow store for larger counts."""
        want = max(self.nbins, int(max_count) + 1)
        nb = min(want, max(self.dense_bins, self.nbins))
        tb = max(self.tbins, int(max_total) + 1)
        if want > nb:
            self._need_overflow()
        if nb != self.nbins:
            if self.K * nb >= (1 << 40):
                raise MemoryError(f"statistics for K={self.K} columns x {nb} dense bins exceed 2^40 cells")
            try:
                new = torch.zeros(self.K, nb, dtype=torch.int64, device=self.device)
            except torch.cuda.OutOfMemoryError as e:
                # (asking the allocator, not mem_get_info: blocks torch holds cached count as free)
                raise MemoryError(
                    f"statistics for K={self.K} columns with {nb} dense count bins take "
                    f"{8 * self.K * nb / 2**30:.1f} GiB on the device, which it does not have; pass a "
                    f"smaller dense_bins to DeviceStats (larger counts then go to the overflow store)"
                ) from e
            new[:, :self.nbins] = self.hist.view(self.K, self.nbins)
            self.hist, self.nbins = new.view(-1), nb
        if tb != self.tbins:
            new = torch.zeros(tb, dtype=torch.int64, device=self.device)
            new[:self.tbins] = self.histV
            self.histV, self.tbins = new, tb

    def _workspace(self, N, X=None):
        need = _lib.lib().dme_stats_workspace_bytes(int(N), self.K) + 256
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off, self._ws.numel() - off

    # -- the statistics pass --------------------------------------------------------------------
    def add_dense(self, X, max_count=None, max_total=None):
        """Adds every row of X (device tensor, uint16 or int32, row-major N x K).  max_count /
        max_total are upper bounds on a cell / a row total; when omitted they are measured with
        one extra streaming pass over X (dme_bounds_dense_*), so pass them when known.  Asynchronous; data errors
        (bounds exceeded, negative cells) surface in check()."""
        if not isinstance(X, torch.Tensor):
            raise TypeError("X must be a torch tensor on the GPU")
        if X.device != self.device:
            raise ValueError(f"X is on {X.device}, statistics on {self.device}")
        if X.dim() != 2 or X.shape[1] != self.K:
            raise ValueError(f"X must be N x {self.K}, got {tuple(X.shape)}")
        if X.shape[0] == 0:
            return self
        if X.stride(1) != 1:
            X = X.contiguous()
        if X.dtype == torch.uint16:
            fn = _lib.lib().dme_stats_dense_u16
        elif X.dtype == torch.int32:
            fn = _lib.lib().dme_stats_dense_i32
        else:
            raise TypeError(f"unsupported dtype {X.dtype}: use uint16 or int32")
        if max_count is None or max_total is None:
            # bounds are measured, not guessed: one extra streaming read of X (dme_bounds_dense_*);
            # avoid it by passing the bounds
            with _on(self.device):
                out = torch.zeros(2, dtype=torch.int64, device=self.device)
                bfn = (_lib.lib().dme_bounds_dense_u16 if X.dtype == torch.uint16
                       else _lib.lib().dme_bounds_dense_i32)
                _lib.check(bfn(X.data_ptr(), X.shape[0], self.K, X.stride(0), _ptr(out), _stream()))
                mc, mt = (int(v) for v in out.cpu())
            max_count = mc if max_count is None else max_count
            max_total = mt if max_total is None else max_total
        self.reserve(max_count, max_total)
        with _on(self.device), _phase("stats"):
            ws, wsb = self._workspace(X.shape[0], X)
            rc = fn(X.data_ptr(), X.shape[0], self.K, X.stride(0), _ptr(self.hist),
                    self.nbins, _ptr(self.histV), self.tbins, _ptr(self.init_acc),
                    _ptr(self.status), _ptr(self.overflow), ws, wsb, _stream())
            if rc == _lib.ERR_UNSUPPORTED:
                # rows wider than the dense kernels take (uint16: 32 768 columns, int32: 16 384)
                return self._add_dense_as_csr(X, max_count, max_total)
            _lib.check(rc)
        return self

    def _add_dense_as_csr(self, X, max_count, max_total):
        """Dense rows of more columns than the dense kernels take: compressed to CSR on the device
        (dme_dense_row_nnz_* -> dme_shard_offsets -> dme_dense_to_csr_*, blocks of at most 2^30 cells)
        and counted by the CSR entry.  One host read per block (the number of non-zero cells)."""
        L = _lib.lib()
        u16 = X.dtype == torch.uint16
        nnz_fn = L.dme_dense_row_nnz_u16 if u16 else L.dme_dense_row_nnz_i32
        csr_fn = L.dme_dense_to_csr_u16 if u16 else L.dme_dense_to_csr_i32
        step = max(1, (1 << 30) // max(self.K, 1))
        with _on(self.device):
            for lo in range(0, X.shape[0], step):
                blk = X[lo:lo + step]
                n = blk.shape[0]
                counts = torch.empty(n, dtype=torch.int64, device=self.device)
                rowptr = torch.empty(n + 1, dtype=torch.int64, device=self.device)
                _lib.check(nnz_fn(blk.data_ptr(), n, self.K, blk.stride(0), _ptr(counts), _stream()))
                _lib.check(L.dme_shard_offsets(_ptr(counts), n, _ptr(rowptr), _stream()))
                nnz = int(rowptr[n].cpu())
                colidx = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.device)[:nnz]
                vals = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.device)[:nnz]
                if nnz:
                    _lib.check(csr_fn(blk.data_ptr(), n, self.K, blk.stride(0), _ptr(rowptr),
                                      _ptr(colidx), _ptr(vals), _stream()))
                self.add_csr(rowptr, colidx, vals, max_count=max_count, max_total=max_total)
        return self

    def add_csr(self, rowptr, colidx, vals, max_count=None, max_total=None):
        """Adds CSR rows (rowptr int64[N+1], colidx int32[nnz], vals int32[nnz], all on device)."""
        for t, dt, nm in ((rowptr, torch.int64, "rowptr"), (colidx, torch.int32, "colidx"),
                          (vals, torch.int32, "vals")):
            if t.dtype != dt or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"{nm} must be a contiguous {dt} tensor on {self.device}")
        N = rowptr.numel() - 1
        if N <= 0:
            return self
        if colidx.numel() != vals.numel():
            raise ValueError("colidx and vals differ in length")
        if max_count is None or max_total is None:  # measured: one pass over rowptr / vals
            with _on(self.device):
                out = torch.zeros(2, dtype=torch.int64, device=self.device)
                _lib.check(_lib.lib().dme_bounds_csr_i32(_ptr(rowptr), _ptr(vals), N, vals.numel(),
                                                         _ptr(out), _stream()))
                mc, mt = (int(v) for v in out.cpu())
            max_count = mc if max_count is None else max_count
            max_total = mt if max_total is None else max_total
        self.reserve(max_count, max_total)
        nnz = colidx.numel()
        with _on(self.device), _phase("stats"):
            need = _lib.lib().dme_stats_csr_workspace_bytes(N, nnz, self.K)
            ws, wsb = 0, 0
            if need > 0:
                if self._csr_ws is None or self._csr_ws.numel() < need + 256:
                    self._csr_ws = None
                    self._csr_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
                off = (-self._csr_ws.data_ptr()) % 256
                ws, wsb = self._csr_ws.data_ptr() + off, self._csr_ws.numel() - off
            _lib.check(_lib.lib().dme_stats_csr_i32(
                _ptr(rowptr), _ptr(colidx), _ptr(vals), N, nnz, self.K, _ptr(self.hist),
                self.nbins, _ptr(self.histV), self.tbins, _ptr(self.init_acc), _ptr(self.status),
                _ptr(self.overflow), ws, wsb, _stream()))
        return self

    def check_async(self):
        """Enqueues the copy of the status words to pinned host memory on the current stream and
        returns at once; a later check() uses it if the stream has been synchronised in between
        (dme_solve does), so the verdict costs no extra device round trip."""
        if getattr(self, "_pinned", None) is None:
            self._pinned = torch.empty(8, dtype=torch.int64, pin_memory=True)
        self._pinned.copy_(self.status, non_blocking=True)
        # no event: recording one costs ~6 us of device time on the stream; check() synchronises
        # the stream instead, which returns at once when dme_solve has already done so
        self._pending = torch.cuda.current_stream()

    def check(self):
        """Raises if the pass met data it could not count (bounds exceeded, negative cells);
        returns rows processed and the maxima seen.  Synchronises unless check_async() ran."""
        ev = getattr(self, "_pending", None)
        if ev is not None:
            ev.synchronize()  # the stream of the copy: idle already when dme_solve ran after it
            self._pending = None
            st = self._pinned.numpy().copy()
        else:
            st = self.status.cpu().numpy()
        return self._verdict(st)

    def _verdict(self, st):
        """The status words of the statistics pass as exceptions / the rows-and-maxima record."""
        if st[6]:
            raise ValueError("CSR input inconsistent: rowptr is negative, decreasing, or spans more "
                             "entries than colidx / vals hold")
        if st[7]:
            raise _lib.DmeError(
                f"{int(st[7])} internal faults in the statistics pass or its all-reduce (a bounded "
                "in-kernel wait expired, or a compact message outgrew the compact_bound promised to "
                "allreduce()): the statistics are incomplete")
        if st[3]:
            raise ValueError(f"{int(st[3])} negative counts (or column indices out of range)")
        if st[1] and self.overflow is not None:
            raise ValueError(
                f"statistics incomplete: the overflow store ({self.overflow_capacity} records) could not "
                f"hold {int(st[1])} cells with a count >= {self.nbins} (largest seen: {int(st[4])}); "
                "pass a larger overflow_capacity or dense_bins to DeviceStats")
        if st[1] or st[2]:
            raise ValueError(
                f"statistics incomplete: {int(st[1])} cells >= {self.nbins} and {int(st[2])} row "
                f"totals >= {self.tbins} (largest seen: count {int(st[4])}, total {int(st[5])}); "
                "max_count / max_total passed to add_* were too small")
        bound = getattr(self, "_rows_bound", None)
        if bound is not None and int(st[0]) > bound:
            raise ValueError(
                f"the ranks counted {int(st[0])} rows, more than the rows_bound={bound} promised to "
                "allreduce(): the histograms travelled in 32-bit words and a bin may have wrapped")
        return {"rows": int(st[0]), "max_count": int(st[4]), "max_total": int(st[5])}

    @property
    def rows(self):
        return int(self.status[0].cpu())

    # -- multi-GPU: rows shard, histograms add ----------------------------------------------------
    def allreduce(self, group=None, same_extents=False, rows_bound=None, compact=None, info=None,
                  force_message=False, compact_bound=None):
        """Sums the partial statistics of all ranks (RCCL over xGMI on GPUs): one integer SUM
        all-reduce, see parallel.allreduce_stats_tensors.  `same_extents=True` when every rank
        passed the same max_count / max_total (no extent agreement round, no host sync);
        `rows_bound` (the same on every rank): an upper bound on the rows of the whole job - below
        2^32 the message travels in 32-bit words, half the bytes; `compact` (default: from 4096
        columns on, and always with an overflow store): only the occupied bins of every column
        travel; `compact_bound` (with same_extents; the same on every rank): a promised upper bound
        on the compact length - the message is sized from it and no host read happens (a longer
        compact form raises status[7], reported by check()); `info`: a dict that receives the message
        size and, in compact form, `compact_words`, the measured length (a caller's next bound).
        After it every rank holds the global statistics and solves redundantly; there is no
        per-iteration communication."""
        from .parallel import allreduce_stats_tensors
        if rows_bound is not None and 0 <= int(rows_bound) < (1 << 32):
            self._rows_bound = int(rows_bound)  # checked against the rows counted, in check()
        with _phase("allreduce"):
            (self.hist, self.histV, self.status, self.init_acc, self.nbins,
             self.tbins) = allreduce_stats_tensors(self.hist, self.histV, self.status, self.init_acc,
                                                   self.K, self.nbins, self.tbins, group, same_extents,
                                                   rows_bound, compact, info, force_message,
                                                   overflow=self, compact_bound=compact_bound)
        return self

    def merge_(self, other):
        """In-place sum with another DeviceStats on the same device (logical shards).  Host-side
        torch arithmetic (off the hot path); the overflow records are appended by dme_overflow_append."""
        if other.K != self.K:
            raise ValueError("K differs")
        if other.nbins > self.dense_bins:
            self.dense_bins = other.nbins  # keep the other's dense cells dense
        self.reserve(other.nbins - 1, other.tbins - 1)
        self.hist.view(self.K, self.nbins)[:, :other.nbins] += other.hist.view(other.K, other.nbins)
        self.histV[:other.tbins] += other.histV
        self.init_acc += other.init_acc
        mx = torch.maximum(self.status[4:6], other.status[4:6])
        self.status += other.status
        self.status[4:6] = mx
        if other.overflow is not None:
            self.overflow_capacity = max(self.overflow_capacity, other.overflow_capacity)
            self._need_overflow()
            with _on(self.device):
                _lib.check(_lib.lib().dme_overflow_append(_ptr(self.overflow), _ptr(other.overflow),
                                                          _ptr(self.status), _stream()))
        return self

    def _overflow_records(self):
        """(columns, values, rows) of the overflow store as int64 host arrays."""
        if self.overflow is None:
            z = np.zeros(0, dtype=np.int64)
            return z, z, z
        head = self.overflow[:2].cpu().numpy()
        used = int(min(head[0], head[1]))
        rec = self.overflow[2:2 + 2 * used].cpu().numpy().reshape(used, 2)
        kv = rec[:, 0].astype(np.uint64)
        cols = (kv & np.uint64(0xFFFFFFFF)).astype(np.int64)
        keep = (cols < self.K) & (rec[:, 1] != 0)
        return cols[keep], (kv >> np.uint64(32)).astype(np.int64)[keep], rec[keep, 1].copy()

    # -- persistence: statistics of row sets computed at different times or places add up -----------
    def save(self, path):
        """Writes the statistics as a .npz file (the histograms' non-zero cells - the dense ones and
        the overflow records alike -, the row-total histogram, the mean accumulator, the status words).
        The reference keeps U/V in memory only; files written here can be loaded and merged (`load`,
        `merge_`) in any order, which gives incremental estimation over data that arrives in pieces.
        Host-side torch / numpy work, off the hot path."""
        self.check()
        h = self.hist.view(self.K, self.nbins)
        nz = torch.nonzero(h)
        ok, ov, on = self._overflow_records()
        np.savez(path, format=np.array([1]), K=np.array([self.K]),
                 nbins=np.array([self.nbins]), tbins=np.array([self.tbins]),
                 cols=np.concatenate([nz[:, 0].to(torch.int32).cpu().numpy(), ok.astype(np.int32)]),
                 bins=np.concatenate([nz[:, 1].to(torch.int32).cpu().numpy(), ov.astype(np.int32)]),
                 counts=np.concatenate([h[nz[:, 0], nz[:, 1]].cpu().numpy(), on]),
                 histV=self.histV.cpu().numpy(), init_acc=self.init_acc.cpu().numpy(),
                 status=self.status.cpu().numpy())

    @classmethod
    def load(cls, path, device=None):
        with np.load(path, allow_pickle=False) as f:
            if int(f["format"][0]) != 1:
                raise ValueError("unknown statistics file format")
            K, nbins, tbins = int(f["K"][0]), int(f["nbins"][0]), int(f["tbins"][0])
            cols_h, bins_h = f["cols"].astype(np.int64), f["bins"].astype(np.int64)
            counts_h = f["counts"].astype(np.int64)
            if cols_h.size and (cols_h.max() >= K or cols_h.min() < 0 or bins_h.min() < 0):
                raise ValueError("statistics file: cell index out of range")
            big = bins_h >= nbins  # cells that were overflow records
            st = cls(K, nbins - 1, tbins - 1, device=device, dense_bins=max(nbins, 2),
                     overflow_capacity=max(int(big.sum()) * 2, OVERFLOW_CAPACITY))
            dev = st.device
            cols = torch.from_numpy(cols_h[~big]).to(dev)
            bins = torch.from_numpy(bins_h[~big]).to(dev)
            st.hist.view(K, st.nbins)[cols, bins] = torch.from_numpy(counts_h[~big]).to(dev)
            st.histV[:tbins] = torch.from_numpy(f["histV"].astype(np.int64)).to(dev)
            st.init_acc.copy_(torch.from_numpy(f["init_acc"].astype(np.float64)).to(dev))
            st.status.copy_(torch.from_numpy(f["status"].astype(np.int64)).to(dev))
            if big.any():
                st._need_overflow()
                n = int(big.sum())
                rec = np.empty((n, 2), dtype=np.int64)
                rec[:, 0] = cols_h[big] | (bins_h[big] << 32)
                rec[:, 1] = counts_h[big]
                st.overflow[2:2 + 2 * n] = torch.from_numpy(rec.reshape(-1)).to(dev)
                st.overflow[0] = n
        return st

    # -- scan -------------------------------------------------------------------------------------
    def finalize(self, ragged=None):
        """Reverse suffix scan -> DeviceUV (the reference's U/V, exact integers + fp64 copies).
        ragged: lay U out with every column at its own length (None: when an overflow store is in
        use or the dense form would take RAGGED_MIN_CELLS cells or more; one host read of the two
        lengths).  Dense-padded otherwise: no host read, which is what keeps estimate_async() on
        BASELINE's K = 1000 shape free of synchronisation."""
        K, dev = self.K, self.device
        if ragged is None:
            ragged = self.overflow is not None or K * self.nbins >= RAGGED_MIN_CELLS
        if not ragged and self.overflow is not None:
            raise ValueError("statistics with an overflow store have no dense-padded form")
        L = _lib.lib()
        capV = max(self.tbins - 1, 1)
        if ragged:
            with _on(dev), _phase("scan"):
                ext = torch.empty(K, dtype=torch.int64, device=dev)
                plan = torch.empty(2 * (K + 1), dtype=torch.int64, device=dev)
                off, slotoff = plan[:K + 1], plan[K + 1:]
                _lib.check(L.dme_shard_extents(_ptr(self.hist), K, self.nbins, _ptr(ext), _stream()))
                if self.overflow is not None:
                    _lib.check(L.dme_overflow_extents(_ptr(self.overflow), K, _ptr(ext), _stream()))
                _lib.check(L.dme_ragged_plan(_ptr(ext), K, _ptr(off), _ptr(slotoff), _stream()))
                total, nslots = (int(v) for v in plan[[K, 2 * K + 1]].cpu())  # the one host read
                tl = max(total, 1)
                # U (int64) | V | lenU | lenV ; the compact cells are written where U goes and scanned
                # in place (a thread writes U[off + v - 1] from the cells it read at that very index)
                ibuf = torch.empty(tl + capV + K + 1, dtype=torch.int64, device=dev)
                U, V = ibuf[:tl], ibuf[tl:tl + capV]
                lenU, lenV = ibuf[tl + capV:tl + capV + K], ibuf[tl + capV + K:]
                fbuf = torch.empty(tl + capV, dtype=torch.float64, device=dev)
                Uf, Vf = fbuf[:tl], fbuf[tl:]
                slotcol = torch.empty(max(nslots, 1), dtype=torch.int32, device=dev)
                _lib.check(L.dme_shard_compact(_ptr(self.hist), K, self.nbins, _ptr(self.overflow),
                                               _ptr(off), _ptr(U), total, _ptr(self.status), _stream()))
                _lib.check(L.dme_suffix_scan_ragged(
                    _ptr(U), _ptr(off), K, _ptr(self.histV), self.tbins, _ptr(U), _ptr(Uf), _ptr(lenU),
                    _ptr(V), _ptr(Vf), _ptr(lenV), _ptr(slotoff), _ptr(slotcol), _stream()))
            return DeviceUV(K, Uf, 1, lenU, Vf, lenV, U=U, V=V,
                            ragged=(off, slotoff, slotcol, nslots, total))
        ldU = max(self.nbins - 1, 1)
        with _on(dev), _phase("scan"):
            # the scan writes every entry (padding included): no zero fill needed
            ibuf = torch.empty(K * ldU + capV + K + 1, dtype=torch.int64, device=dev)
            U, V = ibuf[:K * ldU], ibuf[K * ldU:K * ldU + capV]
            lenU, lenV = ibuf[K * ldU + capV:K * ldU + capV + K], ibuf[K * ldU + capV + K:]
            fbuf = torch.empty(K * ldU + capV, dtype=torch.float64, device=dev)
            Uf, Vf = fbuf[:K * ldU], fbuf[K * ldU:]
            _lib.check(L.dme_suffix_scan(
                _ptr(self.hist), K, self.nbins, _ptr(self.histV), self.tbins, _ptr(U), _ptr(Uf),
                ldU, _ptr(lenU), _ptr(V), _ptr(Vf), _ptr(lenV), _stream()))
        return DeviceUV(K, Uf, ldU, lenU, Vf, lenV, U=U, V=V)

    def estimate_async(self, iterations, hyperprior=0.0):
        """The CLI's tail (DE/fdp.py:104-113) enqueued in one go, without waiting: suffix scan, initial
        alphas from the mean accumulators, the Newton loop (dme_suffix_scan + dme_solve_enqueue).
        Returns a PendingEstimate; its result() waits, raises what check() / solve() would raise and
        returns the solver's record plus `info` (rows, maxima).  The host is free to enqueue the next
        job's statistics pass meanwhile, so the GPU does not idle between jobs."""
        uv = self.finalize()
        return PendingEstimate(self, uv, int(iterations), float(hyperprior))

    def cli_init_alphas(self):
        """DE/fdp.py:104-108 on the device: init_acc / (sum + 0.01) + 0.01."""
        with _on(self.device):
            a0 = torch.empty(self.K, dtype=torch.float64, device=self.device)
            _lib.check(_lib.lib().dme_cli_init_alphas(_ptr(self.init_acc), self.K, _ptr(a0),
                                                      _stream()))
        return a0


class PendingEstimate:
    """A solve that has been enqueued (DeviceStats.estimate_async)."""

    def __init__(self, stats, uv, iterations, hyperprior):
        self.stats, self.uv = stats, uv
        self.iterations, self.hyperprior = iterations, hyperprior
        dev, K = stats.device, stats.K
        with _on(dev):
            self.alpha = torch.empty(K, dtype=torch.float64, device=dev)
            self.cap = min(max(iterations, 1), TRACE_CAP)
            # result[16] int64 | trace[cap][4]
            self.buf = torch.empty(16 + self.cap * 4, dtype=torch.float64, device=dev)
            self.host = torch.zeros(16, dtype=torch.int64, pin_memory=True)
            self.stream = torch.cuda.current_stream()
            self.launched = 0
            self._done = None
            self._enqueue()

    def _enqueue(self):
        uv, st = self.uv, self.stats
        ws, wsb = uv.workspace()
        n = ctypes.c_int64(0)
        tail = (_ptr(uv.lenU), _ptr(uv.Vf), uv.capV, _ptr(uv.lenV), uv.K,
                _ptr(self.alpha), self.iterations, self.hyperprior, _ptr(st.init_acc), _ptr(st.status),
                self.buf.data_ptr() + 16 * 8, 0, self.cap, self.buf.data_ptr(), self.host.data_ptr(),
                ws, wsb, self.launched, 0, ctypes.addressof(n), self.stream.cuda_stream)
        with _phase("solve"):
            if uv.ragged is not None:
                _lib.check(_lib.lib().dme_solve_enqueue_ragged(_ptr(uv.Uf), uv._rgp(), *tail))
            else:
                _lib.check(_lib.lib().dme_solve_enqueue(_ptr(uv.Uf), uv.ldU, *tail))
        self.launched += n.value

    def result(self):
        if self._done is not None:
            return self._done
        with _on(self.stats.device):
            while True:
                self.stream.synchronize()
                if int(self.host[6]):
                    break
                self._enqueue()  # the default batch fell short: more launches, another wait
            res = self.host.tolist()
            # data errors first: statistics that are incomplete (negative cells, counts beyond the
            # bounds, a malformed rowptr) often make the solver fail too, and they are the cause
            info = self.stats._verdict(res[8:16])
            _lib.check(_lib.lib().dme_solve_finish(self.host.data_ptr(), self.uv.K, self.uv.shape_key))
            out = SolveResult(self.alpha, res[:8], self.buf[16:].view(self.cap, 4), None, self.cap,
                              self.uv.K)
            out["info"] = info
            out["uv"] = self.uv
            out["stats"] = self.stats
        self._done = out
        return out


class WeightedStats:
    """fp64 value histograms of rows that carry a weight (appendRow(row, w), w != 1):
    dme_stats_dense_weighted_i32 / dme_suffix_scan_weighted.

    C > 1 keeps C histograms filled from ONE pass over the rows, each row carrying C weights
    (dme_stats_dense_weighted_batch_i32): the C component statistics of one E-step of the mixture
    EM (CPT/dirichletMixtureEstimation.py:128-133), which the reference builds with C appendRow
    calls per row."""

    def __init__(self, K, device=None, C=1):
        self.device = require_gpu(device)
        self.K = int(K)
        self.C = int(C)
        if not 1 <= self.C <= 64:
            raise ValueError("C must be in 1..64")
        self.nbins, self.tbins = 2, 2
        self._ws = None  # workspace of the statistics pass, kept between calls
        self._alloc()

    def _alloc(self):
        d = self.device
        self.histW = torch.zeros(self.C * self.K * self.nbins, dtype=torch.float64, device=d)
        self.histVW = torch.zeros(self.C * self.tbins, dtype=torch.float64, device=d)
        self.maxcol = torch.zeros(self.K, dtype=torch.int64, device=d)
        self.maxtot = torch.zeros(1, dtype=torch.int64, device=d)
        self.status = torch.zeros(8, dtype=torch.int64, device=d)

    def reserve(self, max_count, max_total):
        nb, tb = max(self.nbins, int(max_count) + 1), max(self.tbins, int(max_total) + 1)
        if nb != self.nbins:
            new = torch.zeros(self.C * self.K, nb, dtype=torch.float64, device=self.device)
            new[:, :self.nbins] = self.histW.view(self.C * self.K, self.nbins)
            self.histW, self.nbins = new.view(-1), nb
        if tb != self.tbins:
            new = torch.zeros(self.C, tb, dtype=torch.float64, device=self.device)
            new[:, :self.tbins] = self.histVW.view(self.C, self.tbins)
            self.histVW, self.tbins = new.view(-1), tb

    def add_dense(self, X, weights, max_count, max_total):
        """X: int32 device tensor N x K (contiguous rows); weights: float64 device tensor [N]
        (C == 1) or [N, C]."""
        if X.dtype != torch.int32 or X.dim() != 2 or X.shape[1] != self.K or X.stride(1) != 1:
            raise ValueError("weighted rows must be an int32 N x K device tensor")
        if weights.dtype != torch.float64 or weights.numel() != X.shape[0] * self.C:
            raise ValueError("weights must be float64, %d per row" % self.C)
        if weights.device != self.device or X.device != self.device:
            raise ValueError(f"rows and weights must be on {self.device}")
        weights = weights.contiguous()
        self.reserve(max_count, max_total)
        with _on(self.device):
            L = _lib.lib()
            # the workgroups' images of their bins (added up by a second kernel); small inputs need none
            need = int(L.dme_stats_weighted_workspace_bytes(self.K, self.C, self.nbins, self.tbins))
            if X.shape[0] < 8192:
                need = 0
            if need and (self._ws is None or self._ws.numel() < need):
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(L.dme_stats_dense_weighted_batch_ws_i32(
                X.data_ptr(), _ptr(weights), self.C, X.shape[0], self.K, X.stride(0),
                _ptr(self.histW), self.nbins, _ptr(self.histVW), self.tbins, _ptr(self.maxcol),
                _ptr(self.maxtot), _ptr(self.status), _ptr(self._ws) if need else None, need, _stream()))

    def check(self):
        st = self.status.cpu().numpy()
        if st[3]:
            raise ValueError(f"{int(st[3])} negative counts")
        if st[1] or st[2]:
            raise ValueError("weighted statistics incomplete: bounds passed to add_dense too small")

    def add_into(self, uv, c=0):
        """Adds the weighted suffix sums of weight column c into a DeviceUV (whose U/V stop being
        integers)."""
        if not 0 <= c < self.C:
            raise IndexError("weight column out of range")
        if uv.ragged is not None:
            raise ValueError("weighted statistics add into a dense-padded DeviceUV (finalize(ragged=False))")
        if uv.ldU < self.nbins - 1 or uv.capV < self.tbins - 1:
            raise ValueError("DeviceUV too small for the weighted statistics")
        with _on(self.device):
            _lib.check(_lib.lib().dme_suffix_scan_weighted(
                self.histW.data_ptr() + 8 * c * self.K * self.nbins, self.K, self.nbins,
                self.histVW.data_ptr() + 8 * c * self.tbins, self.tbins,
                _ptr(self.maxcol), _ptr(self.maxtot), _ptr(uv.Uf), uv.ldU, _ptr(uv.lenU),
                _ptr(uv.Vf), uv.capV, _ptr(uv.lenV), _stream()))
        uv.Ui = uv.Vi = None  # no longer exact integers
        return uv

    def to_uv(self, c=0):
        """The statistics of weight column c alone as a DeviceUV (fp64 U/V)."""
        ldU, capV = max(self.nbins - 1, 1), max(self.tbins - 1, 1)
        with _on(self.device):
            buf = torch.zeros(self.K * ldU + capV, dtype=torch.float64, device=self.device)
            lens = torch.zeros(self.K + 1, dtype=torch.int64, device=self.device)
        uv = DeviceUV(self.K, buf[:self.K * ldU], ldU, lens[:self.K], buf[self.K * ldU:], lens[self.K:])
        return self.add_into(uv, c)

    def to_uvs(self):
        """The statistics of all C weight columns as C DeviceUVs from ONE launch and two allocations
        (dme_suffix_scan_weighted_batch): what an E-step hands to the C solves of the M-step."""
        C, K = self.C, self.K
        ldU, capV = max(self.nbins - 1, 1), max(self.tbins - 1, 1)
        with _on(self.device):
            buf = torch.zeros(C * (K * ldU + capV), dtype=torch.float64, device=self.device)
            lens = torch.zeros(C * (K + 1), dtype=torch.int64, device=self.device)
            Uf, Vf = buf[:C * K * ldU], buf[C * K * ldU:]
            lenU, lenV = lens[:C * K], lens[C * K:]
            _lib.check(_lib.lib().dme_suffix_scan_weighted_batch(
                _ptr(self.histW), C, K, self.nbins, _ptr(self.histVW), self.tbins, _ptr(self.maxcol),
                _ptr(self.maxtot), _ptr(Uf), ldU, _ptr(lenU), _ptr(Vf), capV, _ptr(lenV), _stream()))
        uvs = []
        for c in range(C):
            uv = DeviceUV(K, Uf[c * K * ldU:(c + 1) * K * ldU], ldU, lenU[c * K:(c + 1) * K],
                          Vf[c * capV:(c + 1) * capV], lenV[c:c + 1])
            uv.Ui = uv.Vi = None
            uvs.append(uv)
        return uvs


def predicted_step(hconst, hdiag, grad, alphas=None, device=None):
    """getPredictedStep (alphas None) / getPredictedStepAlt on the device (DE/dme.py:67-80)."""
    device = require_gpu(device)
    with _on(device):
        hd = torch.as_tensor(np.asarray(hdiag, dtype=np.float64)).to(device)
        g = torch.as_tensor(np.asarray(grad, dtype=np.float64)).to(device)
        K = g.numel()
        if hd.numel() != K:
            raise ValueError("hDiag and gradient differ in length")
        step = torch.empty(K, dtype=torch.float64, device=device)
        status = torch.zeros(1, dtype=torch.int64, device=device)
        if alphas is None:
            _lib.check(_lib.lib().dme_predicted_step(float(hconst), _ptr(hd), _ptr(g), K,
                                                     _ptr(step), _ptr(status), _stream()))
        else:
            a = torch.as_tensor(np.asarray(alphas, dtype=np.float64)).to(device)
            _lib.check(_lib.lib().dme_predicted_step_alt(float(hconst), _ptr(hd), _ptr(g), _ptr(a),
                                                         K, _ptr(step), _ptr(status), _stream()))
        _lib.check(int(status.cpu()[0]))
        return step.cpu().numpy()


def synth_dirmult(N, K, alpha, totals, seed, dtype=torch.uint16, row_offset=0, device=None,
                  out=None):
    """Synthetic DirichletMultinomial(totals[i], alpha) rows generated on the device
    (dme_synth_dirmult).  totals: int32 device tensor [N].  Returns an N x K tensor."""
    device = require_gpu(device)
    with _on(device):
        alpha = np.asarray(alpha, dtype=np.float64)
        cdf = torch.from_numpy(np.cumsum(alpha)).to(device)
        totals = totals.to(device=device, dtype=torch.int32).contiguous()
        max_total = int(totals.max()) if N else 0
        X = out if out is not None else torch.empty((N, K), dtype=dtype, device=device)
        eb = X.element_size()
        nb = _lib.lib().dme_synth_scratch_bytes(max_total)
        scratch = torch.empty(nb, dtype=torch.uint8, device=device)
        _lib.check(_lib.lib().dme_synth_dirmult(
            X.data_ptr(), eb, N, K, X.stride(0), _ptr(totals), max_total, _ptr(cdf), int(seed),
            int(row_offset), _ptr(scratch), nb, _stream()))
        torch.cuda.current_stream().synchronize()
        del scratch
    return X
