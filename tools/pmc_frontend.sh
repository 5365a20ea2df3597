#!/bin/bash
# Collect PMC counters for the front-end kernels (separate passes; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (on the GPU box, via gpurun): bash tools/pmc_frontend.sh <outdir> [frames]
set -e
OUT=${1:-gpurun_out/pmc}
FR=${2:-200}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-ba > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
[ -n "$NOFETCH" ] || run fetch FETCH_SIZE
[ -n "$NOFETCH" ] || run write WRITE_SIZE
echo done
