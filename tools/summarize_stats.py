#!/usr/bin/env python3
"""Print rocprofv3 --stats kernel rows (calls, total ms, avg us), our kernels first."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        rows.append((name[:60], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:lim]:
    print(f"{r[0]:60s} {r[1]:7d} {r[2]:10.3f} {r[3]:10.2f} {r[4]:6.2f}")
