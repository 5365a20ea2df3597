#!/bin/bash
# Build libvus_hip.so with several FAST tile shapes and bench each (run on the GPU box).
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for cfg in "128 24" "128 16" "128 12" "128 36" "64 36" "256 12" "256 16"; do
  set -- $cfg
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVUS_TW=$1 -DVUS_TH=$2 -c frontend.hip -o frontend.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o -o libvus_hip.so
  echo "TW=$1 TH=$2: $(cd $GRAFT_REPO_ROOT && python bench.py --frames 500 --steps 3 --warmup 1 --no-cpu-baseline --no-ba 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["stage_ms"])')"
done
