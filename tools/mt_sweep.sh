#!/bin/bash
cd $GRAFT_REPO_ROOT/visual-underwater-slam_amd/csrc
for m in 1024 512 256 2048; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVUS_MT=$m -c frontend.hip -o frontend.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vus_common.o frontend.o ba.o nav.o -o libvus_hip.so
  echo "MT=$m: $(cd $GRAFT_REPO_ROOT && python bench.py --frames 500 --steps 3 --warmup 1 --no-cpu-baseline --no-ba 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["stage_ms"])')"
done
