#!/bin/bash
# kernel durations + PMC counters of the Schur kernels at configs[2] (tools/schur_prof.py); usage: bash tools/schur_pmc.sh <outfile>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-gpurun_out/schur_pmc.txt}; case $OUT in /*) ;; *) OUT=$ROOT/$OUT;; esac
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_kt -- python3 $ROOT/tools/schur_prof.py > /dev/null 2>&1
python3 $ROOT/tools/summarize_stats.py /tmp/sp_kt 12 | grep -v "at::native\|rocprim\|rocclr" > $OUT
pmc() { name=$1; shift; rm -rf /tmp/sp_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/sp_$name -- python3 $ROOT/tools/schur_prof.py > /dev/null 2>&1 || echo "pass $name failed"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 $ROOT/tools/summarize_pmc.py /tmp/sp_sq1 /tmp/sp_sq2 /tmp/sp_mfma /tmp/sp_tcc /tmp/sp_fetch /tmp/sp_write | grep -A40 "schur_tiles" >> $OUT
