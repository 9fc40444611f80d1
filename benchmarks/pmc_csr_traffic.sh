#!/bin/bash
# HBM traffic of the CSR pass: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over benchmarks/bow_csr.py.
# usage: benchmarks/pmc_csr_traffic.sh ROWS OUTFILE   (from the repo root on the GPU box)
set -o pipefail
ROWS=${1:-12500000}
OUT=$(realpath ${2:-gpurun_out/pmc_csr_traffic.txt})
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_csr
  timeout -s KILL 200 rocprofv3 --pmc $pass --kernel-include-regex "csr" -d /tmp/pmc_csr -o bow -- python3 $ROOT/benchmarks/bow_csr.py --rows $ROWS --steps 2 > /tmp/pmc_csr.json 2>/tmp/pmc_csr.err || { tail -5 /tmp/pmc_csr.err; exit 1; }
  python3 $ROOT/benchmarks/pmcstats.py /tmp/pmc_csr/bow_results.db csr >> $OUT
done
cat $OUT
