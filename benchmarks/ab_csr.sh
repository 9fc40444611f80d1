#!/bin/bash
# A/B of CSR-pass builds: benchmarks/ab_csr.sh ROWS OUT lib1.so lib2.so ...   (kernel times by rocprofv3)
set -o pipefail
ROWS=$1; OUT=$(realpath $2); shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for lib in "$@"; do
  rm -rf /tmp/ab_csr
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "csr" -d /tmp/ab_csr -o bow -- python3 $ROOT/benchmarks/bow_csr.py --rows $ROWS --steps 5 --lib $ROOT/$lib > /tmp/ab_csr.json 2>/tmp/ab_csr.err || { tail -5 /tmp/ab_csr.err; exit 1; }
  echo "== $lib" >> $OUT
  python3 -c "import json;d=json.load(open('/tmp/ab_csr.json'));print('stats_ms',d['stats_ms'],'min',d['stats_ms_min'],'GB/s',d['achieved_GBps'])" >> $OUT
  python3 $ROOT/benchmarks/kstats.py /tmp/ab_csr/bow_results.db csr >> $OUT
done
cat $OUT
