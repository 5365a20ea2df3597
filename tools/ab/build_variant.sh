#!/bin/bash
# Build a variant of the library for same-box A/B timing (no GPU needed):
#   tools/ab/build_variant.sh NAME "-DSOME_SWITCH"  ->  tools/ab/libvus_n_NAME.so (plain) and libvus_t_NAME.so (-DVUS_TIMING)
# Only ba.hip is rebuilt; the other objects come from visual-underwater-slam_amd/csrc (run make there first).
# The .so files are git-ignored and travel to the GPU box with gpurun; delete them when the experiment is over.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
cd "$HERE/../../visual-underwater-slam_amd/csrc"
name=$1; flags=$2
OFFLOAD=$(make -s print-offload)          # the one place that names the target: csrc/Makefile
mkdir -p /tmp/tb_$name
rm -f "$HERE/libvus_n_$name.so" "$HERE/libvus_t_$name.so"     # a failed build must not leave an older binary to be timed
pids=()
for kind in n t; do
  extra=""; [ $kind = t ] && extra="-DVUS_TIMING"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC $OFFLOAD -Wno-unused-function $extra $flags -c ba.hip -o /tmp/tb_$name/ba_$kind.o \
    && /opt/rocm/bin/hipcc -shared -fPIC $OFFLOAD vus_common.o frontend.o /tmp/tb_$name/ba_$kind.o nav.o pack.o -o "$HERE/libvus_${kind}_$name.so" ) &
  pids+=($!)
done
for pid in "${pids[@]}"; do
  wait "$pid" || { echo "build_variant.sh: building variant '$name' failed" >&2; exit 1; }
done
