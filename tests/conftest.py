import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        fx = json.load(f)
    xpath = os.path.join(GOLDEN, name + "_X.npz")
    if "rows" in fx:
        fx["X"] = np.asarray(fx["rows"], dtype=np.int32).reshape(fx["N"], fx["K"])
    elif os.path.exists(xpath):
        fx["X"] = np.load(xpath, allow_pickle=False)["X"]
    return fx


GOLDEN_CASES = ["test_csv", "test_txt", "one_doublesided", "flipcoins_1500", "heavy_k20",
                "dirmult_k50", "dirmult_k1000", "zero_column"]


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle
    _oracle.build()
    return _oracle
