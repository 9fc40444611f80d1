#!/bin/bash
# Two rocprofv3 --pmc passes over benchmarks/bow_csr.py; prints per-kernel counter sums of the CSR kernels.
# usage: benchmarks/pmc_csr.sh ROWS OUTFILE   (run from the repo root on the GPU box)
set -o pipefail
ROWS=${1:-12500000}
OUT=$(realpath ${2:-gpurun_out/pmc_csr.txt})
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES"; do
  rm -rf /tmp/pmc_csr
  timeout -s KILL 200 rocprofv3 --pmc $pass --kernel-include-regex "csr" -d /tmp/pmc_csr -o bow -- python3 $ROOT/benchmarks/bow_csr.py --rows $ROWS --steps 2 > /tmp/pmc_csr.json 2>/tmp/pmc_csr.err || { tail -5 /tmp/pmc_csr.err; exit 1; }
  python3 $ROOT/benchmarks/pmcstats.py /tmp/pmc_csr/bow_results.db csr >> $OUT
done
cat $OUT
