#!/bin/bash
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for ut in 96 48; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVUS_UT=$ut -c ba.hip -o ba.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o -o libvus_hip.so
  cd /tmp && rm -rf /tmp/prof_ba && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -- python3 $GRAFT_REPO_ROOT/tools/ba_profile.py > /tmp/ba_prof.log 2>&1
  echo "UT=$ut"; python3 $GRAFT_REPO_ROOT/tools/summarize_stats.py /tmp/prof_ba 24 | grep chol
  cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
done
