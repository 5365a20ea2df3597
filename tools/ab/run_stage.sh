#!/bin/bash
# On the GPU box: stage times of each plain variant (tools/band_modes_probe.py, mode 3), alternating, ROUNDS times.
# usage: bash tools/ab/run_stage.sh TAG ROUNDS NAME...
tag=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    VUS_HIP_LIB=$PWD/tools/ab/libvus_n_$v.so timeout -k 10 200 python tools/band_modes_probe.py 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', d['stage_ms'], 'lm_ms', round(1e3 * d['lm_s'], 2))" >> gpurun_out/ab_$tag.log || exit 1
  done
done
cat gpurun_out/ab_$tag.log
