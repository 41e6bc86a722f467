#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch (xrft kernels only)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "xrft::" not in k:
                continue
            k = k.split("(")[0].replace("xrft::", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
