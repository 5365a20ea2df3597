#!/bin/bash
# Front-end twin of build_variant.sh: tools/ab/build_frontend_variant.sh NAME "-DSWITCH ..." -> tools/ab/libvus_fe_NAME.so
# (frontend.hip rebuilt with the switches, the other objects from visual-underwater-slam_amd/csrc; select it with VUS_HIP_LIB).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
cd "$HERE/../../visual-underwater-slam_amd/csrc"
name=$1; flags=$2
OFFLOAD=$(make -s print-offload)
mkdir -p /tmp/tb_fe_$name
rm -f "$HERE/libvus_fe_$name.so"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC $OFFLOAD -Wno-unused-function $(make -s print-frontend-flags) $flags -c frontend.hip -o /tmp/tb_fe_$name/frontend.o
/opt/rocm/bin/hipcc -shared -fPIC $OFFLOAD vus_common.o /tmp/tb_fe_$name/frontend.o ba.o nav.o pack.o -o "$HERE/libvus_fe_$name.so"
/opt/rocm/bin/hipcc -O3 -std=c++17 $OFFLOAD $(make -s print-frontend-flags) $flags -S --cuda-device-only frontend.hip -o /tmp/tb_fe_$name/frontend.s 2>/dev/null
grep -A30 "amdhsa_kernel _ZN12_GLOBAL__N_116fast_tile_kernelILb0ELb1ELb1EEE" /tmp/tb_fe_$name/frontend.s | grep -E "group_segment|next_free_vgpr|scratch" | tr -s '\t\n' ' '; echo " <- $name"
