#!/bin/bash
# PMC counters of orient_rbrief_kernel (separate passes, --pmc with --kernel-trace only): issue / wait split, LDS, and the
# L1 (TCP) -> L2 (TCC) request traffic of its patch gathers.   usage (GPU box): bash tools/pmc_orient.sh <out.txt> [frames]
OUT=$(realpath -m ${1:-gpurun_out/pmc_orient.txt})
FR=${2:-200}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; rm -rf /tmp/pmco_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmco_$name -- python3 $ROOT/bench.py --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-pyramid > /tmp/pmco_$name.log 2>&1 || echo "pass $name failed: $(tail -2 /tmp/pmco_$name.log)"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc tcp2 TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pmc tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc tcc2 TCC_READ_sum TCC_EA0_RDREQ_sum
pmc ta1 TA_BUSY_avr TA_TA_BUSY_sum
pmc fetch FETCH_SIZE
python3 $ROOT/tools/summarize_pmc.py /tmp/pmco_sq1 /tmp/pmco_sq2 /tmp/pmco_tcp1 /tmp/pmco_tcp2 /tmp/pmco_tcc1 /tmp/pmco_tcc2 /tmp/pmco_ta1 /tmp/pmco_fetch | awk '/^[a-z_]/ {p = ($0 ~ /orient_rbrief/)} p' > $OUT
echo "(bench.py --frames $FR: $((2*FR)) images of 1280x720, 2000 keypoints each, per launch)" >> $OUT
