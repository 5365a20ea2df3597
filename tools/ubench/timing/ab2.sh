#!/bin/bash
# usage: ab2.sh tag rounds variants...   band_solve / lm of each non-timing variant, alternating, on one box
tag=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    VUS_HIP_LIB=$PWD/tools/ubench/timing/libvus_n_$v.so timeout -k 10 200 python tools/band_modes_probe.py 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', 'band_solve', d['stage_ms']['band_solve'], 'lm_ms', 1e3 * d['lm_s'], 'err', d['final_error'])" >> gpurun_out/ab_$tag.log || exit 1
  done
done
cat gpurun_out/ab_$tag.log
