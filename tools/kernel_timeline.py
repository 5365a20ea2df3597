#!/usr/bin/env python3
"""Timeline of the kernels of the LAST linear solve in a rocprofv3 --kernel-trace csv directory (tools/ba_profile.py):
start offset, duration, gap to the previous kernel's end, stream -- where the band solve's time outside its kernels is.
usage: python tools/kernel_timeline.py DIR [first_kernel_substring]"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:44], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "lin_points"
starts = [i for i, r in enumerate(rows) if first in r[2]]
seg = rows[starts[-1]:]
t0, prev_end = seg[0][0], seg[0][0]
print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'stream':>6s}  kernel")
for s, e, name, q in seg:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f} {q:>6s}  {name}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
