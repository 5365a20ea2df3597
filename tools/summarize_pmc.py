#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "at::native" in k or "rocclr" in k:
                continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "")
            k = re.sub(r"\(.*", "", k)
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
