#!/bin/bash
# A/B of dense-pass builds by kernel: benchmarks/ab_dense.sh SHAPES OUT dir1 dir2 ...  (a directory that holds benchmarks/dense_shapes.py and a package, or another build of the library)
set -o pipefail
SHAPES=$1; OUT=$(realpath $2); shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for d in "$@"; do
  rm -rf /tmp/ab_dense
  case $d in *.so) PROG="$ROOT/benchmarks/dense_shapes.py --lib $ROOT/$d";; *) PROG="$ROOT/$d/benchmarks/dense_shapes.py";; esac
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/ab_dense -o ds -- python3 $PROG --shapes $SHAPES > /tmp/ab_dense.txt 2>/tmp/ab_dense.err || { tail -5 /tmp/ab_dense.err; exit 1; }
  echo "== $d" >> $OUT
  grep "^|" /tmp/ab_dense.txt >> $OUT
  python3 $ROOT/benchmarks/kstats.py /tmp/ab_dense/ds_results.db | grep -E "stats_|hist_reduce|scan" >> $OUT
done
cat $OUT
