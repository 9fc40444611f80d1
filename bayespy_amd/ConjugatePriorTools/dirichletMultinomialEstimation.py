"""Drop-in for ConjugatePriorTools/dirichletMultinomialEstimation.py ("CPT/dme.py") of
maxsklar/BayesPy: the Dirichlet-multinomial estimator with the (Beta, W) conjugate hyperprior.

Same names, argument order and defaults as the reference; every function computes on the GPU
through libdme_hip.so (dme_eval_cpt / dme_solve_cpt), digamma and trigamma included.  The
statistics container and the step formulas are those of the DirichletEstimation drop-in.
"""
import logging

from .. import device as _dev
from ..dirichletMultinomialEstimation import (  # noqa: F401  (re-exported, identical in CPT/dme.py)
    CompressedRowData, _check_positive_for_log, _uv_of, getPredictedStep, getPredictedStepAlt,
    sqVectorSize)

__all__ = [
    "digamma", "trigamma", "dirichLogProb", "priorGradient", "priorHessianConst",
    "priorHessianDiag", "getPredictedStep", "getPredictedStepAlt", "getTotalLoss",
    "predictStepUsingHessian", "predictStepLogSpace", "testTrialPriors", "sqVectorSize",
    "CompressedRowData", "findDirichletPriors",
]


def _none(x):
    return x is None


def digamma(x):                                                              # CPT/dme.py:15
    import scipy.special as mathExtra
    return float(mathExtra.psi(x))


def trigamma(x):                                                             # CPT/dme.py:16
    import scipy.special as mathExtra
    return float(mathExtra.polygamma(1, x))


#Find the log probability that we see a certain set of data give our prior.   (CPT/dme.py:20-39)
def dirichLogProb(priorList, data, Beta=None, W=None):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0, Beta=Beta, W=W)[0]


#Gives the derivative with respect to the prior.                               (CPT/dme.py:42-66)
def priorGradient(priorList, data, Beta=None, W=None):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0, Beta=Beta, W=W)[2].tolist()


# The hessian is the sum of a diagonal matrix and a constant-value matrix.      (CPT/dme.py:70-87)
def priorHessianConst(priorList, data, W=None):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0, W=W)[1]


def priorHessianDiag(priorList, data, W=None):
    uv = _uv_of(data)
    _check_positive_for_log(priorList, uv)
    return uv.evaluate(priorList, 0.0, W=W)[3].tolist()


# the loss drops the prior terms, as the reference's does                       (CPT/dme.py:109-110)
def getTotalLoss(trialPriors, data, Beta=None, W=None):
    return -1 * dirichLogProb(trialPriors, data, Beta=None, W=None)


def predictStepUsingHessian(gradient, priors, data, W=None):                  # CPT/dme.py:112-115
    uv = _uv_of(data)
    _check_positive_for_log(priors, uv)
    _, hc, _, hd, _ = uv.evaluate(priors, 0.0, W=W)
    return getPredictedStep(hc, hd, gradient)


def predictStepLogSpace(gradient, priors, data, W=None):                      # CPT/dme.py:117-120
    uv = _uv_of(data)
    _check_positive_for_log(priors, uv)
    _, hc, _, hd, _ = uv.evaluate(priors, 0.0, W=W)
    return getPredictedStepAlt(hc, hd, gradient, priors)


# Returns the loss, or inf for a non-positive alpha                              (CPT/dme.py:124-129)
def testTrialPriors(trialPriors, data, Beta=None, W=None):
    for alpha in trialPriors:
        if alpha <= 0:
            return float("inf")
    return getTotalLoss(trialPriors, data, Beta, W)


def findDirichletPriors(data, initAlphas, iterations, Beta=None, W=None):
    """CPT/dme.py:156-219 on the device (dme_solve_cpt): gradient and Hessian carry Beta and W, the
    loss that accepts or rejects a step does not (:109-110)."""
    uv = _uv_of(data)
    want_log = logging.getLogger().isEnabledFor(logging.DEBUG)
    out = uv.solve(initAlphas, iterations, 0.0, trace_alphas=want_log, Beta=Beta, W=W)
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
            return initAlphas
    except TypeError:
        pass
    return priors
