#!/usr/bin/env python3
"""Where the runtime's copy kernels sit in a rocprofv3 --kernel-trace run: for every __amd_rocclr_copyBuffer launch its duration and the
kernels before / after it on the same queue, as a histogram of (previous, next) pairs.  Usage: trace_copies.py DIR"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:48],
                     r.get("Queue_Id", "")))
rows.sort()
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
hist = collections.Counter()
dur = collections.defaultdict(float)
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if "copyBuffer" in r[2] or "fill" in r[2].lower():
            key = (r[2][:24], rs[i - 1][2] if i else "-", rs[i + 1][2] if i + 1 < len(rs) else "-")
            hist[key] += 1
            dur[key] += (r[1] - r[0]) / 1e3
for key, n in hist.most_common(30):
    print(f"{n:5d} x {dur[key] / n:7.1f} us  {key[0]:24s} after {key[1]:48s} before {key[2]}")
