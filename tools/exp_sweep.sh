#!/bin/bash
# usage: tools/exp_sweep.sh "<flags A>" "<flags B>" ...   (rebuilds ba.o on the GPU box per variant, profiles the C3 solve)
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for fl in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c ba.hip -o ba.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
  cd /tmp && rm -rf /tmp/prof_ba && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -- python3 $GRAFT_REPO_ROOT/tools/ba_profile.py > /tmp/ba_prof.log 2>&1
  echo "== $fl"; tail -1 /tmp/ba_prof.log; python3 $GRAFT_REPO_ROOT/tools/summarize_stats.py /tmp/prof_ba 24 | grep -E "chol|schur_blocks"
  cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
done
