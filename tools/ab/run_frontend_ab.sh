#!/bin/bash
# Same-box A/B of front-end library variants (tools/ab/build_frontend_variant.sh): bench.py front-end leg per variant.
for name in "$@"; do
  VUS_HIP_LIB=$PWD/tools/ab/libvus_fe_$name.so timeout -k 10 200 python bench.py --no-ba --no-cpu-baseline > gpurun_out/ab_fe_$name.json 2> gpurun_out/ab_fe_$name.err || { echo "$name failed"; tail -3 gpurun_out/ab_fe_$name.err; exit 1; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_fe_{sys.argv[1]}.json"))
print(f"{sys.argv[1]:10s} {d['value']:10.1f} frames/s  fast_detect {d['stage_ms']['fast_detect']:.3f} ms  parts {d['fast_detect_parts_ms']}")
PY
done
