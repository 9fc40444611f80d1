"""CPU ORACLE — test infrastructure only (never imported by the shipped package).

ctypes front-end for oracle/dme_oracle.c, the loop-faithful C restatement of the reference's
Dirichlet-multinomial hot path (DE/dme.py, DE/fdp.py — see the C file for file:line citations).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity is PINNED: tests/test_oracle_golden.py checks it bit-for-bit against fixtures produced by
the reference itself (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdme_oracle.so")

OK = 0
ZERO_DIVISION = -1
CAPACITY = -2
MATH_DOMAIN = -3

EXIT_NAMES = {0: None, 1: "A", 2: "B", 3: "C"}

_lib = None


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "dme_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_sq_vector_size.restype = ctypes.c_double
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


_i64p = lambda a: _p(a, ctypes.c_int64)
_f64p = lambda a: _p(a, ctypes.c_double)


def _raise(rc):
    if rc == OK:
        return
    if rc == ZERO_DIVISION:
        raise ZeroDivisionError("float division by zero")
    if rc == MATH_DOMAIN:
        raise ValueError("math domain error")
    if rc == CAPACITY:
        raise MemoryError("oracle capacity exceeded")
    raise RuntimeError(f"oracle error {rc}")


class Stats:
    """Ragged U (list of arrays), V (array) — int64 from appendRow with weight 1, float64 once
    weighted rows were added — and the init accumulator (float64[K])."""

    def __init__(self, K, U, V, init_acc):
        self.K = K
        self.U = U
        self.V = V
        self.init_acc = init_acc

    def flat(self):
        off = np.zeros(self.K + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(u) for u in self.U])
        Uflat = (np.concatenate(self.U).astype(np.float64) if off[-1] > 0
                 else np.zeros(1, dtype=np.float64))
        return np.ascontiguousarray(Uflat), off, np.ascontiguousarray(self.V, dtype=np.float64)

    def cli_init_alphas(self):
        a0 = np.zeros(self.K, dtype=np.float64)
        acc = np.ascontiguousarray(self.init_acc, dtype=np.float64)
        lib().oracle_cli_init_alphas(_f64p(acc), ctypes.c_int64(self.K), _f64p(a0))
        return a0


def stats_from_lists(U, V):
    """U/V as the reference holds them (ints, or floats once weighted rows were appended)."""
    K = len(U)
    return Stats(K, [np.asarray(u) for u in U], np.asarray(V), np.zeros(K))


def append_rows_weighted(X, weights, stats=None):
    """appendRow(row, w_i) for every row, literal loops in C; optionally on top of `stats`."""
    X = np.ascontiguousarray(X, dtype=np.int32)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    N, K = X.shape
    capU = max(int(X.max()) if X.size else 0, 1)
    capV = max(int(X.sum(axis=1, dtype=np.int64).max()) if X.size else 0, 1)
    if stats is not None:
        capU = max(capU, max((len(u) for u in stats.U), default=0))
        capV = max(capV, len(stats.V))
    U = np.zeros((K, capU), dtype=np.float64)
    lenU = np.zeros(K, dtype=np.int64)
    V = np.zeros(capV, dtype=np.float64)
    lenV = np.zeros(1, dtype=np.int64)
    if stats is not None:
        for k, u in enumerate(stats.U):
            U[k, :len(u)] = u
            lenU[k] = len(u)
        V[:len(stats.V)] = stats.V
        lenV[0] = len(stats.V)
    rc = lib().oracle_append_rows_weighted_i32(
        _p(X, ctypes.c_int32), _f64p(w), ctypes.c_int64(N), ctypes.c_int64(K), ctypes.c_int64(K),
        _f64p(U), ctypes.c_int64(capU), _i64p(lenU), _f64p(V), ctypes.c_int64(capV), _i64p(lenV))
    _raise(rc)
    return Stats(K, [U[k, :lenU[k]].copy() for k in range(K)], V[:lenV[0]].copy(), np.zeros(K))


def append_rows(X, capU=None, capV=None):
    """appendRow for every row of a dense matrix (uint16 or int32), literal loops in C."""
    X = np.ascontiguousarray(X)
    if X.dtype == np.uint16:
        fn = lib().oracle_append_rows_u16
        ct = ctypes.c_uint16
    else:
        X = np.ascontiguousarray(X, dtype=np.int32)
        fn = lib().oracle_append_rows_i32
        ct = ctypes.c_int32
    N, K = X.shape
    if capU is None:
        capU = int(X.max()) if X.size else 0
    if capV is None:
        capV = int(X.sum(axis=1, dtype=np.int64).max()) if X.size else 0
    capU = max(int(capU), 1)
    capV = max(int(capV), 1)
    U = np.zeros((K, capU), dtype=np.int64)
    lenU = np.zeros(K, dtype=np.int64)
    V = np.zeros(capV, dtype=np.int64)
    lenV = np.zeros(1, dtype=np.int64)
    acc = np.zeros(K, dtype=np.float64)
    rc = fn(_p(X, ct), ctypes.c_int64(N), ctypes.c_int64(K), ctypes.c_int64(K), _i64p(U),
            ctypes.c_int64(capU), _i64p(lenU), _i64p(V), ctypes.c_int64(capV), _i64p(lenV),
            _f64p(acc))
    _raise(rc)
    return Stats(K, [U[k, :lenU[k]].copy() for k in range(K)], V[:lenV[0]].copy(), acc)


def append_rows_csr(rowptr, colidx, vals, K, capU=None, capV=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.int32)
    N = len(rowptr) - 1
    if capU is None:
        capU = int(vals.max()) if vals.size else 0
    if capV is None:
        tot = np.add.reduceat(np.concatenate([vals.astype(np.int64), [0]]), rowptr[:-1]) \
            if N else np.zeros(0)
        tot = np.where(np.diff(rowptr) == 0, 0, tot) if N else tot
        capV = int(tot.max()) if N else 0
    capU = max(int(capU), 1)
    capV = max(int(capV), 1)
    U = np.zeros((K, capU), dtype=np.int64)
    lenU = np.zeros(K, dtype=np.int64)
    V = np.zeros(capV, dtype=np.int64)
    lenV = np.zeros(1, dtype=np.int64)
    acc = np.zeros(K, dtype=np.float64)
    rc = lib().oracle_append_rows_csr_i32(
        _i64p(rowptr), _p(colidx, ctypes.c_int32), _p(vals, ctypes.c_int32), ctypes.c_int64(N),
        ctypes.c_int64(K), _i64p(U), ctypes.c_int64(capU), _i64p(lenU), _i64p(V),
        ctypes.c_int64(capV), _i64p(lenV), _f64p(acc))
    _raise(rc)
    return Stats(K, [U[k, :lenU[k]].copy() for k in range(K)], V[:lenV[0]].copy(), acc)


def _args(stats, alpha):
    Uflat, off, V = stats.flat()
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    assert a.shape == (stats.K,)
    return Uflat, off, V, a


def logprob(alpha, stats, hyperprior=0.0):
    Uflat, off, V, a = _args(stats, alpha)
    out = ctypes.c_double()
    _raise(lib().oracle_logprob(_f64p(a), ctypes.c_int64(stats.K), _f64p(Uflat), _i64p(off),
                                _f64p(V), ctypes.c_int64(len(V)), ctypes.c_double(hyperprior),
                                ctypes.byref(out)))
    return out.value


def total_loss(alpha, stats, hyperprior=0.0):
    Uflat, off, V, a = _args(stats, alpha)
    out = ctypes.c_double()
    _raise(lib().oracle_total_loss(_f64p(a), ctypes.c_int64(stats.K), _f64p(Uflat), _i64p(off),
                                   _f64p(V), ctypes.c_int64(len(V)),
                                   ctypes.c_double(hyperprior), ctypes.byref(out)))
    return out.value


def gradient(alpha, stats, hyperprior=0.0):
    Uflat, off, V, a = _args(stats, alpha)
    g = np.zeros(stats.K)
    _raise(lib().oracle_gradient(_f64p(a), ctypes.c_int64(stats.K), _f64p(Uflat), _i64p(off),
                                 _f64p(V), ctypes.c_int64(len(V)), ctypes.c_double(hyperprior),
                                 _f64p(g)))
    return g


def hconst(alpha, stats):
    Uflat, off, V, a = _args(stats, alpha)
    out = ctypes.c_double()
    _raise(lib().oracle_hconst(_f64p(a), ctypes.c_int64(stats.K), _f64p(V),
                               ctypes.c_int64(len(V)), ctypes.byref(out)))
    return out.value


def hdiag(alpha, stats):
    Uflat, off, V, a = _args(stats, alpha)
    h = np.zeros(stats.K)
    _raise(lib().oracle_hdiag(_f64p(a), ctypes.c_int64(stats.K), _f64p(Uflat), _i64p(off),
                              _f64p(h)))
    return h


def predicted_step(hc, hd, grad):
    hd = np.ascontiguousarray(hd, dtype=np.float64)
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    s = np.zeros(len(grad))
    _raise(lib().oracle_predicted_step(ctypes.c_double(hc), _f64p(hd), _f64p(grad),
                                       ctypes.c_int64(len(grad)), _f64p(s)))
    return s


def predicted_step_alt(hc, hd, grad, alphas):
    hd = np.ascontiguousarray(hd, dtype=np.float64)
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    alphas = np.ascontiguousarray(alphas, dtype=np.float64)
    s = np.zeros(len(grad))
    _raise(lib().oracle_predicted_step_alt(ctypes.c_double(hc), _f64p(hd), _f64p(grad),
                                           _f64p(alphas), ctypes.c_int64(len(grad)), _f64p(s)))
    return s


def sq_vector_size(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    return lib().oracle_sq_vector_size(_f64p(v), ctypes.c_int64(len(v)))


def find_priors(stats, init_alphas, iterations, hyperprior=0.0, eval_cap=4096):
    """findDirichletPriors (DE/dme.py:126-190).  Returns a dict with the returned priors, the
    per-iteration trace the reference logs, the exit reason and every evaluated loss."""
    Uflat, off, V, a = _args(stats, init_alphas)
    a = a.copy()
    K = stats.K
    cap = max(int(iterations), 1)
    t_loss = np.zeros(cap)
    t_gsq = np.zeros(cap)
    t_alpha = np.zeros((cap, K))
    ev = np.zeros(eval_cap)
    n_iters = ctypes.c_int64()
    n_evals = ctypes.c_int64()
    reason = ctypes.c_int()
    rc = lib().oracle_find_priors(
        _f64p(Uflat), _i64p(off), _f64p(V), ctypes.c_int64(K), ctypes.c_int64(len(V)), _f64p(a),
        ctypes.c_int64(iterations), ctypes.c_double(hyperprior), _f64p(t_loss), _f64p(t_gsq),
        _f64p(t_alpha), ctypes.c_int64(cap), ctypes.byref(n_iters), ctypes.byref(n_evals),
        ctypes.byref(reason), _f64p(ev), ctypes.c_int64(eval_cap))
    _raise(rc)
    n = n_iters.value
    return {
        "alpha": a,
        "n_iters": n,
        "n_evals": n_evals.value,
        "exit": EXIT_NAMES[reason.value],
        "trace_loss": t_loss[:n].copy(),
        "trace_gradsq": t_gsq[:n].copy(),
        "trace_alpha": t_alpha[:n].copy(),
        "eval_loss": ev[:min(n_evals.value, eval_cap)].copy(),
    }
