"""Drop-in for the reference's `dirichletMultinomialEstimation` module, backed by HIP kernels.

Same names, argument meaning and error behaviour as
DirichletEstimation/dirichletMultinomialEstimation.py ("DE/dme.py") of maxsklar/BayesPy; the
NumPy port's spelling `dirichletLogProb` (NumpyPort/dirichletMultinomialEstimation.py:19) is
exported as well.  Every function computes on the GPU through libdme_hip.so; there is no CPU path
(a missing GPU or library raises).

`data` may be this module's CompressedRowData or any object exposing the reference's protocol
(.K, .U = K ragged sequences, .V = sequence), e.g. one built by other code.
"""
import logging

import numpy as np

from . import device as _dev

__all__ = [
    "dirichLogProb", "dirichletLogProb", "priorGradient", "priorHessianConst", "priorHessianDiag",
    "getPredictedStep", "getPredictedStepAlt", "getTotalLoss", "predictStepUsingHessian",
    "predictStepLogSpace", "testTrialPriors", "sqVectorSize", "CompressedRowData",
    "findDirichletPriors",
]


class CompressedRowData:
    """DE/dme.py:106-123.  `appendRow(row, weight)` buffers rows on the host; the statistics
    themselves (.U, .V) are built on the GPU when first needed: one pass of the dense statistics
    kernel over the buffered rows, then the reverse suffix scan.  `appendMatrix` / `appendDevice`
    take whole matrices (the fast path)."""

    def __init__(self, K, device=None):
        self.K = K
        self._pending = []       # host rows not yet on the device (lists of K ints)
        self._excess = []        # sum of the entries beyond column K of over-long rows
        self._wpending = []      # rows appended with a weight != 1 ...
        self._wweights = []      # ... and their weights
        self._wstats = None      # WeightedStats
        self._stats = None       # DeviceStats
        self._uv = None          # DeviceUV cache
        self._lists = None       # (U, V) host cache
        self._device = device

    # -- reference protocol ---------------------------------------------------------------------
    def appendRow(self, row, weight):
        if (len(row) != self.K):
            logging.error("row must have K=" + str(self.K) + " counts")  # DE/dme.py:114
        if len(row) < self.K:
            raise IndexError("list index out of range")  # row[k] at DE/dme.py:117
        r = [int(x) for x in row]
        if weight != 1:
            # U[k][j] += weight, V[j] += weight (DE/dme.py:119, :123): kept in fp64 histograms
            if len(r) > self.K and sum(r[self.K:]) != 0:
                raise NotImplementedError("over-long rows with a non-unit weight")
            self._wpending.append(r[:self.K])
            self._wweights.append(float(weight))
            self._uv = None
            self._lists = None
            return
        # an over-long row still counts in full towards sum(row) (DE/dme.py:121)
        self._excess.append(sum(r[self.K:]))
        self._pending.append(r[:self.K])
        self._uv = None
        self._lists = None

    @property
    def U(self):
        return self._host_lists()[0]

    @property
    def V(self):
        return self._host_lists()[1]

    # -- bulk entries (no reference counterpart; what the CLI and the benchmarks use) -------------
    def appendMatrix(self, X):
        """Appends every row of a host matrix (N x K, non-negative integers)."""
        import torch
        X = np.ascontiguousarray(X)
        if X.ndim != 2 or X.shape[1] != self.K:
            raise ValueError(f"expected N x {self.K}")
        if X.dtype not in (np.uint16, np.int32):
            if X.size and (X.min() < -2**31 or X.max() >= 2**31):
                raise OverflowError("counts do not fit int32")
            X = X.astype(np.int32, copy=False)
        self._flush()
        st = self._ensure_stats()
        # the largest count and row total are measured on the device after the upload (a pass over
        # 40M cells costs numpy ~30 ms on one host core, the GPU well under one)
        Xd = torch.from_numpy(X).to(st.device)
        if X.dtype == np.int32 and X.size:
            tot = Xd.sum(dim=1, dtype=torch.int64)
            if int(tot.max()) >= 2**31:
                raise OverflowError("row total does not fit int32")
        st.add_dense(Xd)
        self._uv = None
        self._lists = None

    def appendDevice(self, X, max_count=None, max_total=None):
        """Appends every row of a device tensor (uint16 / int32, N x K) without leaving the GPU."""
        self._flush()
        self._ensure_stats().add_dense(X, max_count=max_count, max_total=max_total)
        self._uv = None
        self._lists = None

    # -- internals ------------------------------------------------------------------------------
    def _ensure_stats(self):
        if self._stats is None:
            self._stats = _dev.DeviceStats(self.K, device=self._device)
        return self._stats

    def _flush_weighted(self):
        if not self._wpending:
            return
        import torch
        X = np.asarray(self._wpending, dtype=np.int64).reshape(len(self._wpending), self.K)
        w = np.asarray(self._wweights, dtype=np.float64)
        self._wpending, self._wweights = [], []
        if X.size and np.abs(X).max() >= 2**31:
            raise OverflowError("counts do not fit int32")
        st = self._ensure_stats()
        if self._wstats is None:
            self._wstats = _dev.WeightedStats(self.K, device=st.device)
        self._wstats.add_dense(torch.from_numpy(X.astype(np.int32)).to(st.device),
                               torch.from_numpy(w).to(st.device),
                               max_count=int(X.max()) if X.size else 0,
                               max_total=int(X.sum(axis=1).max()) if X.size else 0)

    def _flush(self):
        self._flush_weighted()
        if not self._pending:
            return
        import torch
        X = np.asarray(self._pending, dtype=np.int64).reshape(len(self._pending), self.K)
        ex = np.asarray(self._excess, dtype=np.int64)
        self._pending, self._excess = [], []
        if X.size and (np.abs(X).max() >= 2**31 or np.abs(ex).max() >= 2**31):
            raise OverflowError("counts do not fit int32")
        st = self._ensure_stats()
        if ex.any():
            # rows longer than K: the extra entries count towards the row total only; run them as
            # a hidden column K and drop that column's histogram afterwards
            Xe = np.concatenate([X, ex[:, None]], axis=1).astype(np.int32)
            tmp = _dev.DeviceStats(self.K + 1, device=st.device)
            tmp.add_dense(torch.from_numpy(Xe).to(st.device), max_count=int(Xe.max()),
                          max_total=int(Xe.sum(axis=1, dtype=np.int64).max()))
            tmp.check()
            cut = _dev.DeviceStats(self.K, tmp.nbins - 1, tmp.tbins - 1, device=st.device,
                                   dense_bins=tmp.nbins, overflow_capacity=tmp.overflow_capacity)
            cut.overflow = tmp.overflow  # (records of the hidden column K are skipped downstream)
            cut.hist = tmp.hist.view(self.K + 1, tmp.nbins)[:self.K].contiguous().view(-1)
            cut.histV = tmp.histV
            cut.init_acc = tmp.init_acc[:self.K].contiguous()
            cut.status = tmp.status
            st.merge_(cut)
        else:
            X32 = X.astype(np.int32)
            st.add_dense(torch.from_numpy(X32).to(st.device), max_count=int(X32.max()),
                         max_total=int(X32.sum(axis=1, dtype=np.int64).max()))

    def _device_uv(self):
        self._flush()
        if self._uv is None:
            st = self._ensure_stats()
            st.check()
            if self._wstats is not None:
                self._wstats.check()
                st.reserve(self._wstats.nbins - 1, self._wstats.tbins - 1)
            # weighted suffix sums add into a dense-padded U
            self._uv = st.finalize(ragged=False if self._wstats is not None else None)
            if self._wstats is not None:
                self._wstats.add_into(self._uv)
        return self._uv

    def _host_lists(self):
        if self._lists is None or self._pending or self._wpending:
            self._lists = self._device_uv().to_lists()
        return self._lists

    # -- persistence and merging (no reference counterpart: its U/V live in memory only) ----------
    def saveStatistics(self, path):
        """Writes the integer statistics of the rows appended so far (DeviceStats.save)."""
        if self._wpending or self._wstats is not None:
            raise NotImplementedError("weighted rows are not part of the statistics file")
        self.deviceStats().save(path)

    @classmethod
    def loadStatistics(cls, path, device=None):
        st = _dev.DeviceStats.load(path, device=device)
        obj = cls(st.K, device=device)
        obj._stats = st
        return obj

    def merge(self, other):
        """Adds the rows counted in `other` (same K): statistics of disjoint row sets add exactly."""
        if other.K != self.K:
            raise ValueError("K differs")
        if other._wpending or other._wstats is not None:
            raise NotImplementedError("merging weighted rows")
        self.deviceStats().merge_(other.deviceStats())
        self._uv = None
        self._lists = None
        return self

    def deviceStats(self):
        """The underlying DeviceStats (histograms, init accumulator), rows flushed."""
        self._flush()
        return self._ensure_stats()


def _uv_of(data):
    """DeviceUV for `data`: cached for our own class, uploaded for foreign .U/.V objects."""
    if isinstance(data, CompressedRowData):
        return data._device_uv()
    if isinstance(data, _dev.DeviceUV):
        return data
    K = data.K
    U = data.U
    if len(U) != K:
        raise ValueError("data.U must hold K sequences")
    return _dev.DeviceUV.from_lists(U, data.V)


def _check_positive_for_log(priorList, uv):
    # math.log raises ValueError for x <= 0 (DE/dme.py:21, :25); the device path would return NaN
    for a in priorList:
        if a <= 0:
            raise ValueError("math domain error")


#Find the log probability that we see a certain set of data give our prior.   (DE/dme.py:16-29)
def dirichLogProb(priorList, data, hyperprior):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, hyperprior)[0]


dirichletLogProb = dirichLogProb  # NumpyPort spelling


#Gives the derivative with respect to the prior.                               (DE/dme.py:32-48)
def priorGradient(priorList, data, hyperprior=0):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, hyperprior)[2].tolist()


# The hessian is the sum of a diagonal matrix and a constant-value matrix.      (DE/dme.py:52-61)
def priorHessianConst(priorList, data):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0)[1]


def priorHessianDiag(priorList, data):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0)[3].tolist()


# Sherman-Morrison step for H = diag(hDiag) + hConst * 1 1^T                    (DE/dme.py:67-72)
def getPredictedStep(hConst, hDiag, gradient):
    return _dev.predicted_step(hConst, hDiag, gradient).tolist()


# Uses the diagonal hessian on the log-alpha values                              (DE/dme.py:76-80)
def getPredictedStepAlt(hConst, hDiag, gradient, alphas):
    return _dev.predicted_step(hConst, hDiag, gradient, alphas=alphas).tolist()


def getTotalLoss(trialPriors, data, hyperprior=0):                            # DE/dme.py:83-84
    return -1 * dirichLogProb(trialPriors, data, hyperprior)


def predictStepUsingHessian(gradient, priors, data):                          # DE/dme.py:86-89
    uv = _uv_of(data)
    _check_positive_for_log(priors, uv)
    _, hc, _, hd, _ = uv.evaluate(priors, 0.0)
    return getPredictedStep(hc, hd, gradient)


def predictStepLogSpace(gradient, priors, data):                              # DE/dme.py:91-94
    uv = _uv_of(data)
    _check_positive_for_log(priors, uv)
    _, hc, _, hd, _ = uv.evaluate(priors, 0.0)
    return getPredictedStepAlt(hc, hd, gradient, priors)


# Returns the loss, or inf for a non-positive alpha                              (DE/dme.py:97-102)
def testTrialPriors(trialPriors, data, hyperprior=0):
    for alpha in trialPriors:
        if alpha <= 0:
            return float("inf")
    return getTotalLoss(trialPriors, data, hyperprior)


def sqVectorSize(v): return sum(x ** 2 for x in v)                            # DE/dme.py:104


def findDirichletPriors(data, initAlphas, iterations, hyperprior=0):
    """DE/dme.py:126-190 — the whole Newton loop runs on the device (dme_solve); the debug lines
    the reference logs per iteration (:143) and at exit (:146, :183, :189) are replayed from the
    device-side trace."""
    uv = _uv_of(data)
    want_log = logging.getLogger().isEnabledFor(logging.DEBUG)
    out = uv.solve(initAlphas, iterations, hyperprior, trace_alphas=want_log)
    if want_log:
        for i in range(out["n_iters"]):
            loss, gsq = out["trace"][i, 0], out["trace"][i, 1]
            logging.debug("Iteration: %s Loss: %s ,Priors: %s, Gradient Size: %s" % (
                i + 1, float(loss), out["trace_alpha"][i].tolist(), float(gsq)))
        msg = _dev.EXIT_MESSAGES.get(out["exit"])
        if msg:
            logging.debug(msg)
    priors = out["alpha"].tolist()
    try:
        if len(initAlphas) == len(priors) and all(float(a) == b for a, b in zip(initAlphas, priors)):
            return initAlphas  # the reference returns the very object when no step was accepted
    except TypeError:
        pass
    return priors
