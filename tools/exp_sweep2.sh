#!/bin/bash
# like exp_sweep.sh but prints only the schur kernels of the fresh profile
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for fl in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c ba.hip -o ba.o 2>&1 | grep error
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
  cd /tmp && rm -rf /tmp/prof_x && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python3 $GRAFT_REPO_ROOT/tools/ba_profile.py > /tmp/ba_prof.log 2>&1
  echo "== $fl"; python3 $GRAFT_REPO_ROOT/tools/summarize_stats.py /tmp/prof_x 24 | grep -E "schur_"
  cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
done
