#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (launches averaged). Usage: pmc_summary.py DIR [filter]"""
import csv, glob, sys, collections, re
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)[:80]
        if flt and flt not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
