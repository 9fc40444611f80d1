#!/bin/bash
# SQ counters of the weighted statistics kernel: two rocprofv3 --pmc passes over benchmarks/weighted_em.py.
# usage: benchmarks/pmc_weighted.sh OUTFILE   (from the repo root on the GPU box)
set -o pipefail
OUT=$(realpath ${1:-gpurun_out/pmc_weighted.txt})
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES"; do
  rm -rf /tmp/pmc_w
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-include-regex "stats_weighted" -d /tmp/pmc_w -o w -- python3 $ROOT/benchmarks/weighted_em.py --steps 2 > /tmp/pmc_w.json 2>/tmp/pmc_w.err || { tail -5 /tmp/pmc_w.err; exit 1; }
  python3 $ROOT/benchmarks/pmcstats.py /tmp/pmc_w/w_results.db stats_weighted >> $OUT
done
cat $OUT
