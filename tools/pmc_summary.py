#!/usr/bin/env python
"""Collect rocprofv3 --pmc CSV outputs (one directory per pass) into one table:
per kernel name, the mean of every counter over its dispatches."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "pass*", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(k, [])
    print(f"== {k[:90]}  dispatches={len(d)//max(1,len(glob.glob(os.path.join(root,'pass*/'))))} avg_us={(sum(d)/len(d)/1e3 if d else 0):.2f}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} mean={sum(v)/len(v):.6g}  n={len(v)}")
