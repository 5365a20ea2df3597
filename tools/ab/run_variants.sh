#!/bin/bash
# On the GPU box (through gpurun, from the repo root): band solve / LM time of each plain variant, alternating, ROUNDS times,
# then the panel-step cycle marks of each timing variant.   usage: bash tools/ab/run_variants.sh TAG ROUNDS NAME...
tag=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    VUS_HIP_LIB=$PWD/tools/ab/libvus_n_$v.so timeout -k 10 200 python tools/band_modes_probe.py 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', 'band_solve', d['stage_ms']['band_solve'], 'lm_ms', round(1e3 * d['lm_s'], 2), 'err', d['final_error'])" >> gpurun_out/ab_$tag.log || exit 1
  done
done
for v in "$@"; do
  [ -f tools/ab/libvus_t_$v.so ] && VUS_HIP_LIB=$PWD/tools/ab/libvus_t_$v.so timeout -k 10 300 python tools/win_timing.py 2>/dev/null | grep '^{' | sed "s/^/$v /" >> gpurun_out/ab_$tag.log
done
cat gpurun_out/ab_$tag.log
