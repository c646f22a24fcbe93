#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv output (counter_collection) per kernel name: mean counter values per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    if "aa" not in k:
        continue
    d = dur.get(k, [0])
    print(k, f"n={len(d)} avg_us={sum(d) / len(d):.1f}")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
