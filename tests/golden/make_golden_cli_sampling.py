#!/usr/bin/env python3
"""Golden vectors for the CLI's -s / -M options (DE/fdp.py:78, :85), which draw from Python's global
`random` stream: the reference CLI itself, run in this container with the stream seeded, on its own
sample file CPT/test.csv.  Writes tests/golden/cli_sampling.json (arguments, seed, the statistics the
load loop left behind, the printed priors).  Run from the repository root; needs /root/reference."""
import io
import json
import os
import random
import runpy
import sys

REF = "/root/reference/DirichletEstimation"
SAMPLE = "/root/reference/ConjugatePriorTools/test.csv"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_sampling.json")


def run(argv, seed):
    sys.path.insert(0, REF)
    old = sys.argv, sys.stdin, sys.stdout
    buf = io.StringIO()
    try:
        sys.argv = [os.path.join(REF, "findDirichletPrior.py")] + argv
        sys.stdin = open(SAMPLE)
        sys.stdout = buf
        random.seed(seed)
        g = runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        sys.stdin.close()
        sys.argv, sys.stdin, sys.stdout = old
        sys.path.remove(REF)
    data = g["dataObj"]
    line = [ln for ln in buf.getvalue().splitlines() if ln.startswith("Final priors:")][-1]
    return {"argv": argv, "seed": seed, "U": [list(u) for u in data.U], "V": list(data.V),
            "rows_kept": int(data.V[0]) if len(data.V) else 0, "stdout": line,
            "priors": json.loads(line.split(":", 1)[1])}


def main():
    import logging
    logging.disable(logging.CRITICAL)
    cases = [run(["-K", "2", "-s", "0.3", "-M", "50"], 7),
             run(["-K", "2", "-s", "0.5"], 11),
             run(["-K", "2", "-M", "20"], 13),
             run(["-K", "2", "-s", "1", "-M", "35", "-H", "0.01"], 17)]
    with open(OUT, "w") as f:
        json.dump({"input": "ConjugatePriorTools/test.csv", "cases": cases}, f)
    for c in cases:
        print(c["argv"], c["seed"], c["rows_kept"], c["stdout"])


if __name__ == "__main__":
    main()
