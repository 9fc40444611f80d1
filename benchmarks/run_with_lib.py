#!/usr/bin/env python3
"""Any benchmark script against another build of the library (A/B comparisons on one box):
  python benchmarks/run_with_lib.py bayespy_amd/libdme_other.so benchmarks/heavy_tail.py [arguments]
Run from the repository root."""
import sys, runpy, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "benchmarks"))
from bayespy_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
