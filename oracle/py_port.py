"""CPU ORACLE — test infrastructure only (never imported by the shipped package).

Loop-faithful CPython restatement of the reference's statistics build, for timing the reference's
own cost model (an interpreted loop per row) on the GPU box, where the reference itself is absent:
CompressedRowData.appendRow (DirichletEstimation/dirichletMultinomialEstimation.py:113-123) and the
CLI's mean accumulator (DirichletEstimation/findDirichletPrior.py:87-89).  Checked against the C
oracle (itself pinned on the reference's golden vectors) in tests/test_oracle_golden.py.
"""


class CompressedRowData:
    def __init__(self, K):                                                   # DE/dme.py:107-111
        self.K = K
        self.V = []
        self.U = []
        for k in range(0, K):
            self.U.append([])

    def appendRow(self, row, weight):                                        # DE/dme.py:113-123
        for k in range(0, self.K):
            for j in range(0, row[k]):
                if len(self.U[k]) == j:
                    self.U[k].append(0)
                self.U[k][j] += weight
        for j in range(0, sum(row)):
            if len(self.V) == j:
                self.V.append(0)
            self.V[j] += weight


def build(rows, K):
    """Rows (lists of K ints) -> (CompressedRowData, mean accumulator) as the CLI loop builds them."""
    priors = [0.0] * K
    data = CompressedRowData(K)
    for row in rows:
        sumData = sum(row)                                                   # DE/fdp.py:87-89
        weightForMean = 1.0 / (1.0 + sumData)
        for i in range(0, K):
            priors[i] += row[i] * weightForMean
        data.appendRow(row, 1)
    return data, priors
