#!/usr/bin/env python3
"""Finding the optimal dirichlet prior from counts — GPU drop-in for the reference CLI.

Same command line, stdin and stdout as DirichletEstimation/findDirichletPrior.py ("DE/fdp.py")
of maxsklar/BayesPy, plus the NumPy port's -d/--delimiter (NumpyPort/findDirichletPrior.py:57):

    cat test.csv | ./findDirichletPrior.py --sampleRate 1

Rows are parsed on the host exactly as the reference does (csv.reader, int()), then handed to the
GPU as one matrix: the statistics pass, the suffix scan, the CLI's initial-alpha formula
(DE/fdp.py:87-89, 104-108) and the whole Newton loop run in HIP kernels.
"""
import os
import sys

if __package__ in (None, ""):  # executed as a script: make the package importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = "bayespy_amd"

import csv
import logging
import random
import time
from optparse import OptionParser

import numpy as np

from . import dirichletMultinomialEstimation as DME
from . import samplingTools as Sample


def _parse_fast(raw, delimiter, K):
    """N x K int32 matrix from delimiter-separated text via dme_parse_counts, or None when the text
    is outside the fast parser's grammar (the caller then runs the reference's row loop)."""
    import ctypes

    from . import _lib
    if isinstance(raw, str):
        raw = raw.encode()
    if not raw or ord(delimiter) >= 128:
        return None
    L = _lib.lib()
    n = L.dme_parse_count_rows(raw, len(raw))
    out = np.empty((max(n, 1), K), dtype=np.int32)
    got = ctypes.c_int64(0)
    rc = L.dme_parse_counts(raw, len(raw), ord(delimiter), K, out.ctypes.data, n,
                            ctypes.addressof(got), min(os.cpu_count() or 1, 16))
    if rc != 0 or got.value != n or n == 0:
        return None
    return out[:n]


_CHUNK_BYTES = 256 << 20  # stdin is parsed and counted in pieces of this size


def _row_loop(lines, dataObj, K, sampleRate, M, delimiter, idx):
    """The reference's row loop: csv.reader + int() per cell, one Bernoulli draw per row for -s,
    random thinning of rows above the -M cap (DE/fdp.py:75-92).  Returns the rows read so far."""
    for fields in csv.reader(lines, delimiter=delimiter):
        idx += 1
        if random.random() < sampleRate:
            counts = [int(f) for f in fields]
            if len(counts) != K:
                logging.error("There are %s categories, but line has %s counts." % (K, len(counts)))
                logging.error("line %s: %s" % (idx, counts))
            while sum(counts) > M:
                counts[Sample.drawCategory(counts)] -= 1
            # the mean accumulator of DE/fdp.py:87-89 is computed by the statistics kernel
            dataObj.appendRow(counts, 1)
        if idx % 1000000 == 0:
            logging.debug("Loading Data: %s rows done" % idx)
    return idx


def load_data(stdin, K, sampleRate, M, delimiter):
    """The load loop of DE/fdp.py:66-93 -> (CompressedRowData, rows read)."""
    csv.field_size_limit(1000000000)
    logging.debug("Loading data")
    dataObj = DME.CompressedRowData(K)

    # Fast path (libdme_hip's multi-threaded parser): every row is kept (-s 1) and uncapped (no -M),
    # so no per-row random draw decides anything.  stdin is read in pieces that end on a line
    # break; each piece is parsed, counted on the GPU and dropped, so memory stays bounded whatever
    # the input size.  From the first piece outside the parser's strict grammar onwards the
    # reference's own loop takes over (it keeps the reference's error behaviour).
    fast = sampleRate >= 1.0 and M == sys.maxsize and len(delimiter) == 1
    if not fast:
        return dataObj, _row_loop(stdin, dataObj, K, sampleRate, M, delimiter, 0)

    binary = hasattr(stdin, "buffer")
    src = stdin.buffer if binary else stdin
    idx = 0
    carry = b"" if binary else ""
    nl = b"\n" if binary else "\n"
    while True:
        block = src.read(_CHUNK_BYTES)
        piece = carry + block
        if block:
            cut = piece.rfind(nl)
            if cut < 0:  # no line break yet: keep reading
                carry = piece
                continue
            piece, carry = piece[:cut + 1], piece[cut + 1:]
        else:
            carry = piece[:0]
        if piece:
            X = _parse_fast(piece, delimiter, K)
            if X is None:
                # hand this piece and everything after it to the reference's loop
                import io
                import itertools
                text = piece.decode() if binary else piece
                rest = carry.decode() if binary else carry
                tail = io.TextIOWrapper(src) if binary else src
                lines = itertools.chain(text.splitlines(True), _lines_after(rest, tail))
                return dataObj, _row_loop(lines, dataObj, K, sampleRate, M, delimiter, idx)
            before = idx
            idx += X.shape[0]
            for done in range((before // 1000000 + 1) * 1000000, idx + 1, 1000000):
                logging.debug("Loading Data: %s rows done" % done)
            dataObj.appendMatrix(X)
        if not block:
            break
    return dataObj, idx


def _lines_after(prefix, stream):
    """Lines of prefix + the rest of stream (prefix is the unfinished line read ahead)."""
    first = True
    for line in stream:
        yield prefix + line if first else line
        first = False
    if first and prefix:
        yield prefix


def main(argv=None, stdin=None, stdout=None):
    stdin = sys.stdin if stdin is None else stdin
    stdout = sys.stdout if stdout is None else stdout
    startTime = time.time()
    parser = OptionParser()
    parser.add_option('-s', '--sampleRate', dest='sampleRate', default='1', help='keep each input row with this probability (1 keeps all)')
    parser.add_option('-K', '--numCategories', dest='K', default='2', help='number of count columns per row')
    parser.add_option('-M', '--maxCountPerRow', dest='M', type=int, default=sys.maxsize, help='rows whose counts sum to more are thinned down to this total')
    parser.add_option("-L", '--loglevel', action="store", dest="loglevel", default='DEBUG', help="logging level of the status messages on stderr (ERROR|WARNING|INFO|DEBUG)")
    parser.add_option('-H', '--hyperPrior', dest='H', default='0', help='scalar hyperprior lambda (0 = plain maximum likelihood)')
    parser.add_option('-i', '--iterations', dest='iterations', default='50', help='upper bound on Newton iterations')
    parser.add_option('-d', '--delimiter', dest='delimiter', default='\t', help='column delimiter of the input (default: tab)')

    (options, args) = parser.parse_args(argv)

    K = int(options.K)
    iterations = int(options.iterations)
    # Set the log level
    log_level = options.loglevel
    numeric_level = getattr(logging, log_level, None)
    if not isinstance(numeric_level, int):
        raise ValueError('Invalid log level: %s' % log_level)
    logging.basicConfig(level=numeric_level)

    logging.debug("K = " + str(K))

    hyperprior = float(options.H)
    logging.debug("Hyperprior (lambda) = " + str(hyperprior))

    dataObj, idx = load_data(stdin, K, float(options.sampleRate), options.M, options.delimiter)

    dataLoadTime = time.time()
    logging.debug("loaded %s records into memory" % idx)
    logging.debug("time to load memory: %s " % (dataLoadTime - startTime))

    # initial priors: DE/fdp.py:104-108 on the device
    stats = dataObj.deviceStats()
    stats.check()
    priors = stats.cli_init_alphas().cpu().numpy().tolist()

    priors = DME.findDirichletPriors(dataObj, priors, iterations, hyperprior)

    print("Final priors: " + str(priors), file=stdout)
    logging.debug("Final average loss: %s" % DME.getTotalLoss(priors, dataObj, hyperprior))

    totalTime = time.time() - dataLoadTime
    logging.debug("Time to calculate: %s" % totalTime)
    return priors


if __name__ == "__main__":
    main()
