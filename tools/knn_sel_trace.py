#!/usr/bin/env python3
"""Where the rows of knn_sel_kernel spend their cycles (diagnosis build -DNSID_KSEL_TRACE, NSID_LIB=<that .so>): cycles of wave 0 of
every workgroup per section of the strip loop, summed over its 32 rows.
Usage: NSID_LIB=neuralsampleid_amd/libnsid_hip_kseltrace.so python tools/knn_sel_trace.py [--k 18 --d 1]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
from neuralsampleid_amd._lib import lib
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--C", type=int, default=64)
ap.add_argument("--k", type=int, default=18); ap.add_argument("--d", type=int, default=1)
a = ap.parse_args()
N = 256
r = torch.randn(a.B * N, a.C, device="cuda").bfloat16()
aff = ops.BNAffine(torch.rand(a.C, device="cuda") + 0.5, torch.randn(a.C, device="cuda") * 0.1)
for _ in range(3):
    ops.knn_graph(r, a.B, N, a.C, a.k, a.d, aff)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.knn_graph(r, a.B, N, a.C, a.k, a.d, aff)
e1.record(); torch.cuda.synchronize()
print(f"knn_sel B={a.B} C={a.C} k={a.k} d={a.d}: {e0.elapsed_time(e1) * 100:.1f} us per launch (back to back)")
buf = torch.zeros(8 * a.B, dtype=torch.int64, device="cuda")
assert lib.nsid_debug_knn_trace(buf.data_ptr()) == 0
ops.knn_graph(r, a.B, N, a.C, a.k, a.d, aff)
torch.cuda.synchronize()
lib.nsid_debug_knn_trace(None)
t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ["(loop head)", "phase A (MFMA strip -> keys)", "rows: bisection", "rows: compaction", "rows: candidate ranks + store", "-", "-"]
tot = t[:, :7].sum(1)
print(f"  cycles of wave 0, median over workgroups; total {np.median(tot):.0f}")
for i, n in enumerate(names):
    print(f"  {n:46s} {np.median(t[:, i]):9.0f}  ({np.median(t[:, i]) / 32:7.0f} per row)")
