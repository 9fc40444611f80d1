#!/usr/bin/env python3
"""bench.py against another build of the library (A/B comparisons on one box):
  python benchmarks/bench_with_lib.py bayespy_amd/libdme_other.so [bench.py arguments]
Prints bench.py's JSON line; run from the repository root."""
import sys, runpy, os
sys.path.insert(0, os.getcwd())
from bayespy_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
