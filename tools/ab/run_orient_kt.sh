#!/bin/bash
# Per-kernel durations of the orientation / descriptor launches for front-end library variants (rocprofv3 kernel trace of the
# front-end leg of bench.py).  usage (GPU box): [KT_PATTERN=fast_tile] bash tools/ab/run_orient_kt.sh NAME...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  rm -rf /tmp/kt_$name
  VUS_HIP_LIB=$ROOT/tools/ab/libvus_fe_$name.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -- python3 $ROOT/bench.py --no-ba --no-cpu-baseline --no-pyramid > /tmp/kt_$name.json 2>/dev/null
  echo "== $name: $(python3 -c "import json;d=json.loads(open('/tmp/kt_$name.json').read().strip().splitlines()[-1]);print(d['value'], d['stage_ms']['orient_rbrief'])")"
  python3 $ROOT/tools/summarize_stats.py /tmp/kt_$name 40 | grep "${KT_PATTERN:-orient}" | cut -c1-120 | head -6
done
