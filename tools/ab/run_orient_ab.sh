#!/bin/bash
# Same-box A/B of front-end library variants on the orientation + rBRIEF stage: bench.py front-end leg per variant,
# two rounds (alternating) so that a drift of the box shows.
mkdir -p gpurun_out
for round in 1 2; do
for name in "$@"; do
  VUS_HIP_LIB=$PWD/tools/ab/libvus_fe_$name.so timeout -k 10 200 python bench.py --no-ba --no-cpu-baseline --no-pyramid > gpurun_out/ab_fe_$name.json 2> gpurun_out/ab_fe_$name.err || { echo "$name failed"; tail -3 gpurun_out/ab_fe_$name.err; exit 1; }
  python - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab_fe_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:10s} {d['value']:10.1f} frames/s  orient_rbrief {d['stage_ms']['orient_rbrief']:.3f} ms  fast {d['stage_ms']['fast_detect']:.3f}  track {d['stage_ms']['hamming_track']:.3f}")
PY
done
done
