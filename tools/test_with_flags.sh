#!/bin/bash
# usage: tools/test_with_flags.sh "<ba.hip flags>" <pytest args...>  -- rebuild ba.o with flags on the GPU box, run pytest
fl="$1"; shift
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c ba.hip -o ba.o 2>&1 | grep error
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
cd $GRAFT_REPO_ROOT && python -m pytest "$@"
