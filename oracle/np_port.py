"""CPU ORACLE — test infrastructure only (never imported by the shipped package).

NumPy-structured restatement of the reference's statistics build, for timing the cost model of the
reference's NumPy port on the GPU box, where the reference itself is absent:
NumpyPort/dirichletMultinomialEstimation.py:100-121 keeps U[k] and V as float64 ndarrays, lengthens
them with np.append when a larger count arrives and adds the weight to a slice per column per row.
The mean accumulator is the CLI's (DirichletEstimation/findDirichletPrior.py:87-89).  Checked
against the C oracle (itself pinned on the reference's golden vectors) in
tests/test_oracle_golden.py.
"""
import numpy as np


class CompressedRowData:
    def __init__(self, K):                                                  # NP/dme.py:101-104
        self.K = K
        self.V = np.zeros(0)
        self.U = [np.zeros(0) for _ in range(K)]

    def appendRow(self, row, weight):                                       # NP/dme.py:106-121
        for k in range(self.K):
            n = row[k]
            have = len(self.U[k])
            if have < n:
                self.U[k] = np.append(self.U[k], np.zeros(n - have))
            self.U[k][:n] += weight
        total = sum(row)
        have = len(self.V)
        if have < total:
            self.V = np.append(self.V, np.zeros(total - have))
        self.V[:total] += weight


def build(rows, K):
    """Rows (lists of K ints) -> (CompressedRowData, mean accumulator) as the CLI loop builds them."""
    priors = [0.0] * K
    data = CompressedRowData(K)
    for row in rows:
        weightForMean = 1.0 / (1.0 + sum(row))                              # DE/fdp.py:87-89
        for i in range(K):
            priors[i] += row[i] * weightForMean
        data.appendRow(row, 1)
    return data, priors
