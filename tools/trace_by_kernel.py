#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace run over the last N steps (delimited by adam_kernel), FULL template names kept.
Usage: trace_by_kernel.py DIR [N]"""
import collections
import csv
import glob
import re
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
wall = (max(r[1] for r in sel) - sel[0][0]) / 1e3 / n
tot = sum(e - s for s, e, _ in sel) / 1e3 / n
print(f"steps {n}: wall/step {wall:.1f} us, sum of kernel durations/step {tot:.1f} us, launches/step {len(sel) / n:.0f}")
acc = collections.defaultdict(lambda: [0, 0])
for s, e, name in sel:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((GemmArgs|WgArgs|.*)\)$", "", name).replace(" ", "")
    acc[name][0] += 1
    acc[name][1] += e - s
for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"  {name[:110]:110s} x{c / n:6.1f} avg {t / c / 1e3:7.1f} us  per-step {t / 1e3 / n:8.1f} us")
