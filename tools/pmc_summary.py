#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 --pmc CSV output (counter_collection.csv files)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "ppoxp" not in name:
                continue
            short = name.split("ppoxp::")[1].split("(")[0]
            if not short.startswith("gae_scan_kernel"):      # the summary / fix-up / plain scans differ only by template arguments
                short = short.split("<")[0]
            short = short.replace(" ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        v = v[len(v) // 3:]  # drop warm-up dispatches
        print(f"{k:36s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
