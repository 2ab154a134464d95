#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/prof_pmc.sh) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r.get("Kernel_Name", "")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
for k in sorted(agg):
    if k.startswith("__amd") or "at::native" in k: continue
    print(k)
    for c in names:
        if c in agg[k]:
            v = agg[k][c]
            print("    %-24s mean %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
