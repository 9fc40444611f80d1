#!/bin/bash
# Round-3 evidence from ONE tree on one MI355X box: bench line, rocprofv3 kernel statistics, PMC traffic
# (tied to the kernel source blob), single-GPU shares of the strong-scaling job through the one-rank RCCL
# collective, the CSR / weighted / heavy-tail / dense-shape side benchmarks.
# usage: benchmarks/r3_final_profiles.sh COMMIT   (from the repo root on the GPU box; writes gpurun_out/r3/)
COMMIT=${1:-unknown}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
step() {  # step NAME SECONDS cmd...: a GPU step under its own limit; a kill or crash ends the script
  local name=$1 secs=$2; shift 2
  echo "== $name"
  timeout -k 10 $secs "$@"
  local rc=$?
  echo "== $name rc=$rc"
  case $rc in 124|137|134|139) echo "stopping: $name was killed or crashed"; exit $rc;; esac
  return $rc
}
# the PMC traffic first: bench.py reads profiles/traffic.json (and reports null when it was measured on other sources)
step traffic 600 bash benchmarks/pmc_bench_traffic.sh $COMMIT $OUT && cp $OUT/traffic.json profiles/traffic.json
step bench 300 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
step kernel_trace 300 bash -c "cd /tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python3 $ROOT/bench.py --no-cpu-baseline > /tmp/kt.json 2>/tmp/kt.err && python3 $ROOT/benchmarks/kstats.py /tmp/kt/b_results.db > $OUT/bench_kernel_stats.txt && cp /tmp/kt.json $OUT/bench_under_rocprof.json"
for rows in 5000000 2500000 1250000; do
  step share_$rows 200 python3 bench.py --rows $rows --force-collective --no-cpu-baseline > $OUT/share_rows_$rows.json 2> $OUT/share_rows_$rows.err
done
step share_10000000 200 python3 bench.py --force-collective --no-cpu-baseline > $OUT/share_rows_10000000.json 2> $OUT/share_rows_10000000.err
step c2 200 python3 bench.py --k 2 --rows 1000000 --max-total 60 --dtype i32 --no-cpu-baseline > $OUT/bench_c2_k2_1M.json 2> $OUT/bench_c2_k2_1M.err
step csr 400 bash benchmarks/ab_csr.sh 12500000 $OUT/csr_bow12p5M_kernel_stats.txt bayespy_amd/libdme_hip.so
step csr_steps 300 python3 benchmarks/bow_csr.py --rows 12500000 --steps 16 --warmup 2 --solve > $OUT/csr_bow12p5M.json 2> $OUT/csr_bow12p5M.err
step csr_traffic 500 bash benchmarks/pmc_csr_traffic.sh 12500000 $OUT/csr_pmc_traffic.txt
step weighted 200 python3 benchmarks/weighted_em.py > $OUT/weighted_em.json 2> $OUT/weighted_em.err
step weighted_pmc 300 bash benchmarks/pmc_weighted.sh $OUT/weighted_pmc_sq.txt
step heavy_tail 300 python3 benchmarks/heavy_tail.py > $OUT/heavy_tail.json 2> $OUT/heavy_tail.err
step roctx 300 bash -c "cd /tmp && rm -rf /tmp/mk && DME_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -d /tmp/mk -o m -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --settle-seconds 0 > /tmp/mk.json 2>/tmp/mk.err; python3 $ROOT/benchmarks/roctx_ranges.py /tmp/mk/m_results.db > $OUT/roctx_ranges.txt 2>&1"
# (SKIP_DENSE=1: without the 33-shape table, which takes ten minutes and only changes with dme_stats.hip)
[ -n "$SKIP_DENSE" ] || step dense_shapes 900 python3 benchmarks/dense_shapes.py > $OUT/dense_shapes.txt 2> $OUT/dense_shapes.err
echo done
