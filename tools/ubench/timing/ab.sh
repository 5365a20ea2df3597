#!/bin/bash
# usage: ab.sh tag variants...   (runs on the GPU box from the repo root)
tag=$1; shift
for v in "$@"; do
  echo "== n_$v" >> gpurun_out/ab_$tag.log
  VUS_HIP_LIB=$PWD/tools/ubench/timing/libvus_n_$v.so timeout -k 10 200 python tools/band_modes_probe.py 3 2>/dev/null | grep '^{' >> gpurun_out/ab_$tag.log || exit 1
done
for v in "$@"; do
  echo "== t_$v" >> gpurun_out/ab_$tag.log
  VUS_HIP_LIB=$PWD/tools/ubench/timing/libvus_t_$v.so timeout -k 10 200 python tools/band_modes_probe.py 3 2>/dev/null | grep '^WT\|^PF' | sort | awk 'NR%4==1' | head -12 >> gpurun_out/ab_$tag.log || exit 1
done
cat gpurun_out/ab_$tag.log
