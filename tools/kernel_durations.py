"""Durations of every dispatch of the kernels whose name contains argv[2], from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?")))
for t, d, g, l in sorted(rows):
    print(f"{d:9.1f} us  grid {g}  lds {l}")
