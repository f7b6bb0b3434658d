#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over the LAST n graph replays: per-kernel time, union busy time, overlap.
Usage: trace_summary.py DIR [--steps N] [--launches-per-step L]"""
import csv, glob, sys, re, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# find adam kernels as step delimiters
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lo, hi = adam[-nsteps - 1] + 1, adam[-1] + 1
sel = rows[lo:hi]
t0, t1 = sel[0][0], max(r[1] for r in sel)
wall = (t1 - t0) / 1e3 / nsteps
# union
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in sel)
print(f"steps {nsteps}: wall/step {wall:.1f} us, union busy/step {busy/1e3/nsteps:.1f} us, sum of kernel durations/step {tot/1e3/nsteps:.1f} us, launches/step {len(sel)/nsteps:.0f}")
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)[:60]
    acc[n][0] += 1; acc[n][1] += e - s
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {n:60s} x{c/nsteps:6.1f} avg {t/c/1e3:7.1f}us  per-step {t/1e3/nsteps:8.1f}us")
