"""List per-call durations of one kernel from a rocprofv3 --kernel-trace CSV directory, grouped by grid size."""
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            g = (r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
            rows[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(g, "calls", len(v), "avg_us %.1f" % (sum(v) / len(v)))
