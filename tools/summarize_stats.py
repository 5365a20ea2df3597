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

# Per launch size: a kernel that the command launches at several sizes (bench.py runs the headline stream of 1000 frames
# AND the small end-to-end / first-call legs) has one average per grid here; the headline launch is the largest grid.
by = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
        grid = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        by.setdefault(name, {}).setdefault(grid, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
multi = [(n, g) for n, g in by.items() if len(g) > 1 and n in {r[0] for r in rows[:lim]}]
if multi:
    print(f"\n{'kernel by launch size (threads in the grid)':60s} {'grid':>10s} {'calls':>7s} {'avg_us':>10s}")
    for n, g in sorted(multi, key=lambda x: -max(sum(v) for v in x[1].values())):
        for grid in sorted(g, reverse=True)[:4]:
            v = g[grid]
            print(f"{n:60s} {grid:10d} {len(v):7d} {sum(v) / len(v) / 1e3:10.2f}")
