#!/usr/bin/env python3
"""The end of one training step in a rocprofv3 --kernel-trace run: the last N launches before the optimiser's adam_kernel, with start / end
relative to the step's first launch, plus the step's wall time and kernel-time sum.  Usage: trace_tail.py DIR [N]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
sel = rows[adam[-2] + 1: adam[-1] + 1]
t0 = sel[0][0]
print(f"launches {len(sel)}, wall {(sel[-1][1] - t0) / 1e3:.1f} us, kernel sum {sum(e - s for s, e, _, _ in sel) / 1e3:.1f} us")
big = [r for r in sel if "wgrad_grouped" in r[2]]
for s, e, name, q in big:
    print(f"  grouped: {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  q{q}")
for s, e, name, q in sel[-n:]:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    print(f"  {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  q{q}  {name[:100]}")
