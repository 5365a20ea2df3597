#!/bin/bash
# PMC counters of the BA kernels at configs[2] (tools/ba_profile.py), separate passes as MI355X_MICROARCH.md prescribes.
# usage (GPU box): bash tools/pmc_ba.sh <outfile>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-gpurun_out/pmc_ba.txt}; case $OUT in /*) ;; *) OUT=$ROOT/$OUT;; esac
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; rm -rf /tmp/pmcba_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcba_$name -- python3 $ROOT/tools/ba_profile.py > /dev/null 2>&1 || echo "pass $name failed"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 $ROOT/tools/summarize_pmc.py /tmp/pmcba_sq1 /tmp/pmcba_sq2 /tmp/pmcba_mfma /tmp/pmcba_tcp /tmp/pmcba_tcc /tmp/pmcba_fetch /tmp/pmcba_write --launch-traffic-json $(dirname $OUT)/traffic_ba.json > $OUT
