#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per (kernel, counter) for our kernels."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(list)
dur = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ctmr" not in k:
            continue
        k = k.split("(")[0].replace("ctmr::", "")
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        if "Start_Timestamp" in row and row["Start_Timestamp"]:
            dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:28s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f} last={v[-1]:16.1f}")
for k, v in sorted(dur.items()):
    print(f"{k:28s} duration_ns mean={sum(v)/len(v):12.0f} n={len(v)}")
