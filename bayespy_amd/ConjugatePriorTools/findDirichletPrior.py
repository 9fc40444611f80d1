#!/usr/bin/env python3
"""Finding the optimal dirichlet prior from counts, with the (Beta, W) hyperprior — GPU drop-in for
ConjugatePriorTools/findDirichletPrior.py ("CPT/fdp.py") of maxsklar/BayesPy.

    cat test.csv | ./findDirichletPrior.py -K 3 -H 0.1,0.1,0.1,2

-H takes K+1 comma-separated values (Beta then W, CPT/fdp.py:59-69); anything else means no
hyperprior, and then an all-zero column is an error (CPT/fdp.py:109-112).
"""
import os
import sys

if __package__ in (None, ""):  # executed as a script: make the package importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    __package__ = "bayespy_amd.ConjugatePriorTools"

import logging
import time
from optparse import OptionParser

from . import dirichletMultinomialEstimation as DME
from ..findDirichletPrior import load_data


def main(argv=None, stdin=None, stdout=None):
    stdin = sys.stdin if stdin is None else stdin
    stdout = sys.stdout if stdout is None else stdout
    startTime = time.time()
    parser = OptionParser()
    parser.add_option('-s', '--sampleRate', dest='sampleRate', default='1', help='Randomly sample this fraction of rows')
    parser.add_option('-K', '--numCategories', dest='K', default='2', help='The number of (tab separated) categories that are being counted')
    parser.add_option('-M', '--maxCountPerRow', dest='M', type=int, default=sys.maxsize, help='The maximum number of the count per row.  Setting this lower increases the running time')
    parser.add_option("-L", '--loglevel', action="store", dest="loglevel", default='DEBUG', help="don't print status messages to stdout")
    parser.add_option('-H', '--hyperPrior', dest='H', default="", help='The hyperprior of the Dirichlet (paper coming soon!) comma separated K+1 values (Beta then W)')
    parser.add_option('-i', '--iterations', dest='iterations', default='50', help='How many iterations to do')

    (options, args) = parser.parse_args(argv)

    K = int(options.K)
    iterations = int(options.iterations)
    #Set the log level
    log_level = options.loglevel
    numeric_level = getattr(logging, log_level, None)
    if not isinstance(numeric_level, int):
        raise ValueError('Invalid log level: %s' % log_level)
    logging.basicConfig(level=numeric_level)

    logging.debug("K = " + str(K))

    W = 0
    Beta = [0] * K
    Hstr = options.H.split(",")
    hasHyperprior = False
    if (len(Hstr) == K + 1):
        for i in range(0, K): Beta[i] = float(Hstr[i])
        W = float(Hstr[K])
        hasHyperprior = True
    else:
        Beta = None
        W = None

    logging.debug("Beta = " + str(Beta))
    logging.debug("W = " + str(W))

    dataObj, idx = load_data(stdin, K, float(options.sampleRate), options.M, '\t')

    dataLoadTime = time.time()
    logging.debug("loaded %s records into memory" % idx)
    logging.debug("time to load memory: %s " % (dataLoadTime - startTime))

    stats = dataObj.deviceStats()
    stats.check()
    # a column whose every row is 0 leaves U[k] empty                        (CPT/fdp.py:109-112)
    allZero = stats.hist.view(K, stats.nbins)[:, 1:].sum(dim=1) == 0
    if not hasHyperprior and bool(allZero.any()):
        raise Exception("You can't have any columns with all 0s, unless you provide a hyperprior (-H)")

    # initial priors: CPT/fdp.py:114-118 on the device
    priors = stats.cli_init_alphas().cpu().numpy().tolist()

    priors = DME.findDirichletPriors(dataObj, priors, iterations, Beta, W)

    print("Final priors: ", priors, file=stdout)
    logging.debug("Final average loss: %s" % DME.getTotalLoss(priors, dataObj, Beta, W))

    totalTime = time.time() - dataLoadTime
    logging.debug("Time to calculate: %s" % totalTime)
    return priors


if __name__ == "__main__":
    main()
