#!/bin/bash
# usage: tools/exp_run.sh "<flags>"  -- rebuild ba.o with flags on the box and run the C3 solve once (no profiler)
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $1 -c ba.hip -o ba.o 2>&1 | grep -E "error" 
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
cd $GRAFT_REPO_ROOT && python3 tools/ba_profile.py 2>&1 | tail -12
