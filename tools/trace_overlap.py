#!/usr/bin/env python3
"""Concurrency profile of a rocprofv3 --kernel-trace run over the last N steps (delimited by adam_kernel): wall, union busy, time with
0 / 1 / 2+ kernels in flight, and the longest idle gaps. Usage: trace_overlap.py DIR [N]"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sel = rows[adam[-n - 1] + 1: adam[-1] + 1]
t0, t1 = sel[0][0], max(r[1] for r in sel)
ev = sorted([(s, 1) for s, e, _ in sel] + [(e, -1) for s, e, _ in sel])
level, last, hist, gaps = 0, t0, {}, []
for t, dlt in ev:
    if t > last:
        hist[min(level, 3)] = hist.get(min(level, 3), 0) + (t - last)
        if level == 0:
            gaps.append(t - last)
    level += dlt
    last = t
wall = (t1 - t0) / 1e3 / n
print(f"steps {n}: wall/step {wall:.1f} us; time with 0 / 1 / 2 / 3+ kernels in flight per step: " +
      " / ".join(f"{hist.get(k, 0) / 1e3 / n:.0f}" for k in range(4)) + " us")
gaps.sort(reverse=True)
print("idle gaps per step:", len(gaps) / n, "longest (us):", [round(g / 1e3, 1) for g in gaps[:8]])
