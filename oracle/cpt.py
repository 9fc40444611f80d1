"""CPU ORACLE — test infrastructure only (never imported by the shipped package).

Restatement of the (Beta, W) conjugate-hyperprior variant of the estimator,
ConjugatePriorTools/dirichletMultinomialEstimation.py ("CPT/dme.py") of maxsklar/BayesPy, with the
reference's own special functions (scipy.special.psi / polygamma, CPT/dme.py:15-16).  Inner sums
are numpy-vectorised, so values agree with the reference to rounding (1e-13), not bit for bit.
PINNED: tests/test_oracle_golden.py::test_cpt_* check it against tests/golden/cpt_hyperprior.json,
which tests/golden/make_golden_cpt.py produced by running the reference module itself.
"""
import math

import numpy as np
import scipy.special as sp


class UV:
    """Ragged U / V of CompressedRowData (CPT/dme.py:136-153) as float arrays."""

    def __init__(self, U, V):
        self.K = len(U)
        self.U = [np.asarray(u, dtype=np.float64) for u in U]
        self.V = np.asarray(V, dtype=np.float64)
        self.iU = [np.arange(len(u), dtype=np.float64) for u in self.U]
        self.iV = np.arange(len(self.V), dtype=np.float64)


def log_prob(a, d, Beta=None, W=None):                                       # CPT/dme.py:20-39
    a = np.asarray(a, dtype=np.float64)
    total = 0.0
    for k in range(d.K):
        total += float(np.sum(d.U[k] * np.log(a[k] + d.iU[k])))
    A = float(np.sum(a))
    total -= float(np.sum(d.V * np.log(A + d.iV)))
    if Beta is not None:
        total -= float(np.dot(a, np.asarray(Beta, dtype=np.float64)[:d.K]))
    if W is not None:
        total += W * math.lgamma(A)
        total -= W * float(np.sum(sp.gammaln(a)))
    return total


def gradient(a, d, Beta=None, W=None):                                       # CPT/dme.py:42-66
    a = np.asarray(a, dtype=np.float64)
    A = float(np.sum(a))
    sub = float(np.sum(d.V / (A + d.iV)))
    g = np.array([float(np.sum(d.U[k] / (a[k] + d.iU[k]))) for k in range(d.K)]) - sub
    if Beta is not None:
        g = g - np.asarray(Beta, dtype=np.float64)[:d.K]
    if W is not None:
        g = g + W * (float(sp.psi(A)) - sp.psi(a))
    return g


def hconst(a, d, W=None):                                                    # CPT/dme.py:70-76
    A = float(np.sum(a))
    t = float(np.sum(d.V / (A + d.iV) ** 2))
    if W is not None:
        t += W * float(sp.polygamma(1, A))
    return t


def hdiag(a, d, W=None):                                                     # CPT/dme.py:78-87
    a = np.asarray(a, dtype=np.float64)
    h = -np.array([float(np.sum(d.U[k] / (a[k] + d.iU[k]) ** 2)) for k in range(d.K)])
    if W is not None:
        h = h - W * sp.polygamma(1, a)
    return h


def step(hc, hd, g):                                                         # CPT/dme.py:93-98
    if np.any(hd == 0) or hc == 0:
        raise ZeroDivisionError("float division by zero")
    b = float(np.sum(g / hd)) / (1.0 / hc + float(np.sum(1.0 / hd)))
    return (b - g) / hd


def step_alt(hc, hd, g, a):                                                  # CPT/dme.py:102-106
    x = g + a * hd
    if np.any(x == 0) or hc == 0:
        raise ZeroDivisionError("float division by zero")
    Z = 1.0 / hc + float(np.sum(a / x))
    S = float(np.sum(a * g / x))
    return (S / Z - g) / x


def total_loss(a, d, Beta=None, W=None):                                     # CPT/dme.py:109-110
    return -log_prob(a, d, None, None)


def trial_loss(a, d):                                                        # CPT/dme.py:124-129
    if np.any(np.asarray(a) <= 0):
        return float("inf")
    return total_loss(a, d)


def find_priors(d, init, iterations, Beta=None, W=None):                     # CPT/dme.py:156-219
    priors = np.asarray(init, dtype=np.float64).copy()
    cur = total_loss(priors, d)
    n_evals, trace = 1, []
    count = 0
    while count < iterations:
        count += 1
        g = gradient(priors, d, Beta, W)
        gsq = float(np.sum(g ** 2))
        trace.append({"loss": cur, "gradSq": gsq, "priors": priors.copy()})
        if gsq < 2 ** -10:
            return {"alpha": priors, "exit": "A", "n_iters": count, "n_evals": n_evals, "trace": trace}
        hc, hd = hconst(priors, d, W), hdiag(priors, d, W)
        trial = priors + step(hc, hd, g)
        loss = trial_loss(trial, d)
        n_evals += 1
        if loss < cur:
            cur, priors = loss, trial
            continue
        with np.errstate(over="ignore"):
            e = np.exp(step_alt(hc, hd, g, priors))
        trial = np.where(np.isfinite(e), priors * e, priors)   # OverflowError -> keep (:197-200)
        loss = trial_loss(trial, d)
        n_evals += 1
        lr = 1.0
        while loss > cur:
            lr *= 0.9
            trial = priors + g * lr
            loss = trial_loss(trial, d)
            n_evals += 1
        if lr < 2 ** -20:
            return {"alpha": priors, "exit": "B", "n_iters": count, "n_evals": n_evals, "trace": trace}
        cur, priors = loss, trial
    return {"alpha": priors, "exit": "C", "n_iters": count, "n_evals": n_evals, "trace": trace}
