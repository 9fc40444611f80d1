#!/usr/bin/env python3
"""Finding the optimal dirichlet prior from counts — GPU drop-in for the reference CLI.

Same command line, stdin and stdout as DirichletEstimation/findDirichletPrior.py ("DE/fdp.py")
of maxsklar/BayesPy, plus the NumPy port's -d/--delimiter (NumpyPort/findDirichletPrior.py:57):

    cat test.csv | ./findDirichletPrior.py --sampleRate 1

stdin is parsed in pieces by libdme_hip's multi-threaded parser (text outside its strict grammar
goes through the reference's own csv.reader + int() loop), each piece handed to the GPU as one
matrix: the statistics pass, the suffix scan, the CLI's initial-alpha formula (DE/fdp.py:87-89,
104-108) and the whole Newton loop run in HIP kernels.  -s and -M keep the reference's semantics and
its use of Python's `random` stream draw for draw (DE/fdp.py:78, :85), so a seeded run selects and
thins the same rows on both paths.
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


def _parse_fast(raw, delimiter, K, keep=None):
    """N x K int32 matrix from delimiter-separated text via dme_parse_counts, or None when the text
    is outside the fast parser's grammar (the caller then runs the reference's row loop).
    keep (uint8 per line): only those lines are parsed (dme_parse_counts_selected)."""
    import ctypes

    from . import _lib
    if isinstance(raw, str):
        raw = raw.encode()
    if not raw or ord(delimiter) >= 128:
        return None
    L = _lib.lib()
    n = L.dme_parse_count_rows(raw, len(raw)) if keep is None else int(np.count_nonzero(keep))
    out = np.empty((max(n, 1), K), dtype=np.int32)
    got = ctypes.c_int64(0)
    threads = min(os.cpu_count() or 1, 16)
    if keep is None:
        rc = L.dme_parse_counts(raw, len(raw), ord(delimiter), K, out.ctypes.data, n,
                                ctypes.addressof(got), threads)
        if n == 0:
            return None
    else:
        rc = L.dme_parse_counts_selected(raw, len(raw), ord(delimiter), K, keep.ctypes.data, keep.size,
                                         out.ctypes.data, n, ctypes.addressof(got), threads)
    if rc != 0 or got.value != n:
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


class _Stream:
    """Python's global `random` stream, many draws at a time.  random.random() is MT19937 with the
    53-bit construction (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, which is also what numpy's legacy
    RandomState.random_sample computes: a RandomState loaded with random.getstate() yields the very
    doubles the next calls of random.random() would, a million of them in milliseconds, and its end
    state goes back into `random`.  Checked against random.random() itself on first use; if the two
    ever disagreed (another generator behind `random`), draws fall back to one call each."""

    _bulk_ok = None

    @classmethod
    def _rs(cls):
        st = random.getstate()
        if st[0] != 3 or len(st[1]) != 625:
            return None, st
        rs = np.random.RandomState()
        rs.set_state(("MT19937", np.array(st[1][:-1], dtype=np.uint32), int(st[1][-1])))
        return rs, st

    @classmethod
    def _check(cls):
        if cls._bulk_ok is None:
            saved = random.getstate()
            try:
                rs, _ = cls._rs()
                ok = rs is not None
                if ok:
                    want = [random.random() for _ in range(700)]  # crosses a state regeneration
                    ok = rs.random_sample(700).tolist() == want
                cls._bulk_ok = bool(ok)
            finally:
                random.setstate(saved)
        return cls._bulk_ok

    @classmethod
    def peek(cls, n):
        """The next n draws WITHOUT consuming them (None when bulk draws are unavailable)."""
        if not cls._check():
            return None
        rs, _ = cls._rs()
        return rs.random_sample(int(n))

    @classmethod
    def advance(cls, n):
        """Consumes n draws."""
        rs, st = cls._rs()
        rs.random_sample(int(n))
        ns = rs.get_state()
        random.setstate((st[0], tuple(int(x) for x in ns[1]) + (int(ns[2]),), st[2]))

    @classmethod
    def take(cls, n):
        """The next n draws, consumed."""
        u = cls.peek(n)
        if u is None:
            return np.fromiter((random.random() for _ in range(n)), dtype=np.float64, count=n)
        cls.advance(n)
        return u


def _select_and_cap(X, sampleRate, M):
    """-s and -M on a parsed piece, with the reference's own draws in the reference's own order
    (DE/fdp.py:78 `if random.random() < sampleRate` once per row, then DE/fdp.py:85
    `while sum(counts) > M: counts[drawCategory(counts)] -= 1` for a kept row, before the next row's
    draw): seeded with random.seed(), the fast path keeps and thins exactly the rows the row loop
    would.  Returns the kept rows as a matrix (rows above the cap thinned; X is reused)."""
    import ctypes

    from . import _lib
    n, K = X.shape
    L = _lib.lib()
    cap = min(int(M), 2**62)
    need = L.dme_cap_draws_needed(X.ctypes.data, n, K, cap)
    u = _Stream.peek(need)
    if u is not None:
        kept, used = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(L.dme_sample_and_cap_i32(X.ctypes.data, n, K, float(sampleRate), cap, u.ctypes.data,
                                            u.size, ctypes.addressof(kept), ctypes.addressof(used)))
        _Stream.advance(used.value)
        return X[:kept.value]
    # no bulk draws: the loop itself, draw by draw
    keep = np.empty(n, dtype=bool)
    over = (X.sum(axis=1, dtype=np.int64) > M).tolist()
    for i in range(n):
        k = random.random() < sampleRate
        keep[i] = k
        if k and over[i]:
            counts = X[i].tolist()
            while sum(counts) > M:
                counts[Sample.drawCategory(counts)] -= 1
            X[i] = counts
    return X[keep]


def load_data(stdin, K, sampleRate, M, delimiter):
    """The load loop of DE/fdp.py:66-93 -> (CompressedRowData, rows read)."""
    csv.field_size_limit(1000000000)
    logging.debug("Loading data")
    dataObj = DME.CompressedRowData(K)

    # Fast path (libdme_hip's multi-threaded parser).  stdin is read in pieces that end on a line
    # break; each piece is parsed, sampled / capped (-s, -M: _select_and_cap), counted on the GPU and
    # dropped, so memory stays bounded whatever the input size.  With -s 1 and no -M no draw decides
    # anything and none is made.  From the first piece outside the parser's strict grammar onwards
    # the reference's own loop takes over (it keeps the reference's error behaviour, and continues
    # the same random stream).
    fast = len(delimiter) == 1
    plain = sampleRate >= 1.0 and M == sys.maxsize
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
            keep = None
            if sampleRate < 1.0 and M == sys.maxsize:
                # -s alone: one draw per line decides, before any field is looked at (DE/fdp.py:78-79)
                from . import _lib
                raw_piece = piece if binary else piece.encode()
                n_lines = _lib.lib().dme_parse_count_rows(raw_piece, len(raw_piece))
                saved = random.getstate()
                keep = (_Stream.take(n_lines) < sampleRate).astype(np.uint8)
            X = _parse_fast(piece, delimiter, K, keep)
            if X is None:
                if keep is not None:
                    random.setstate(saved)  # the row loop below makes this piece's draws itself
                # hand this piece and everything after it to the reference's loop
                import io
                import itertools
                text = piece.decode() if binary else piece
                rest = carry.decode() if binary else carry
                tail = io.TextIOWrapper(src) if binary else src
                lines = itertools.chain(text.splitlines(True), _lines_after(rest, tail))
                return dataObj, _row_loop(lines, dataObj, K, sampleRate, M, delimiter, idx)
            before = idx
            idx += X.shape[0] if keep is None else keep.size
            for done in range((before // 1000000 + 1) * 1000000, idx + 1, 1000000):
                logging.debug("Loading Data: %s rows done" % done)
            if not plain and keep is None:
                X = _select_and_cap(X, sampleRate, M)
            if X.shape[0]:
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
