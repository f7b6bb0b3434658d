#!/usr/bin/env python3
"""Per-clip timeline of one kNN launch (nsid_debug_knn_trace): staging, normalisation and distance/top-k phases.
Usage: python tools/knn_trace.py [--N 64 --C 256 --k 3]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
from neuralsampleid_amd._lib import lib
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--N", type=int, default=64)
ap.add_argument("--C", type=int, default=256); ap.add_argument("--k", type=int, default=3)
a = ap.parse_args()
r = torch.randn(a.B * a.N, a.C, device="cuda").bfloat16()
aff = ops.BNAffine(torch.rand(a.C, device="cuda") + 0.5, torch.randn(a.C, device="cuda") * 0.1)
for _ in range(5):
    ops.knn_graph(r, a.B, a.N, a.C, a.k, 1, aff)
torch.cuda.synchronize()
buf = torch.zeros(4 * a.B, dtype=torch.int64, device="cuda")
assert lib.nsid_debug_knn_trace(buf.data_ptr()) == 0
ops.knn_graph(r, a.B, a.N, a.C, a.k, 1, aff)
torch.cuda.synchronize()
lib.nsid_debug_knn_trace(None)
t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
base = t[:, 0].min()
print(f"kNN B={a.B} N={a.N} C={a.C} k={a.k}: span {(t[:, 3].max() - base) / 100:.2f} us; starts within {(t[:, 0].max() - base) / 100:.2f} us")
for name, i, j in (("stage features", 0, 1), ("normalise", 1, 2), ("distances + top-k", 2, 3), ("whole workgroup", 0, 3)):
    d = (t[:, j] - t[:, i]) / 100
    print(f"  {name:18s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
