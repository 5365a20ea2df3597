#!/bin/bash
# usage: tools/fe_sweep.sh "<flags A>" "<flags B>" ...  (rebuild frontend.o per variant on the GPU box, run the front-end bench)
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for fl in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $fl -c frontend.hip -o frontend.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
  echo "== $fl"
  (cd $GRAFT_REPO_ROOT && python3 bench.py --no-ba --no-cpu-baseline --no-pyramid --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])")
done
