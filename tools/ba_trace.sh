#!/bin/bash
# kernel-trace of tools/ba_profile.py under a few environment settings; prints per-kernel stats (timing experiments)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rm -rf /tmp/prof_$tag; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $ROOT/tools/ba_profile.py > /dev/null 2>&1; echo "== $tag ($*)"; python3 $ROOT/tools/summarize_stats.py /tmp/prof_$tag 12 | grep -v "at::native\|rocprim\|rocclr" | tee $OUT/stats_$tag.txt; }
run default A=1

run fused VUS_BAND_MODE=0

