#!/bin/bash
# HBM traffic of the bench's statistics launch: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# bench.py, then profiles/traffic.json written with the blob of the kernel source and the commit.
# usage: benchmarks/pmc_bench_traffic.sh COMMIT OUTDIR   (from the repo root on the GPU box)
set -o pipefail
COMMIT=${1:-unknown}
OUT=$(realpath ${2:-gpurun_out})
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$pass
  timeout -s KILL 240 rocprofv3 --pmc $pass --kernel-include-regex "stats_dense" -d /tmp/pmc_$pass -o b -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0 > /tmp/pmc_$pass.json 2>/tmp/pmc_$pass.err || { tail -5 /tmp/pmc_$pass.err; exit 1; }
  python3 $ROOT/benchmarks/pmcstats.py /tmp/pmc_$pass/b_results.db stats_dense > $OUT/pmc_${pass}_stats_dense.txt
done
python3 $ROOT/benchmarks/traffic_json.py /tmp/pmc_FETCH_SIZE/b_results.db /tmp/pmc_WRITE_SIZE/b_results.db $COMMIT > $OUT/traffic.json
cat $OUT/traffic.json
