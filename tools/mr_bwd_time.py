#!/usr/bin/env python3
"""mr_bwd_kernel per stage shape: microseconds per launch (back to back).  Usage: python tools/mr_bwd_time.py [--B 256 --k 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=256); ap.add_argument("--k", type=int, default=3); ap.add_argument("--tune", action="append", default=[]); ap.add_argument("--hub", type=float, default=0.0)
a = ap.parse_args()
for kv in a.tune:
    k_, v_ = kv.split("="); ops.set_tuning(k_, int(v_))
for N, C in ((256, 64), (128, 128), (64, 256), (32, 512)):
    for dt in (torch.bfloat16, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(N)
        r = torch.randn(a.B * N, C, device="cuda", generator=g).to(dt)
        idx = torch.randint(0, N, (a.B, N, a.k), device="cuda", generator=g, dtype=torch.int32)
        if a.hub > 0:
            idx = torch.where(torch.rand(a.B, N, a.k, device="cuda", generator=g) < a.hub, torch.zeros_like(idx), idx)
        u, am = ops.mr_aggregate_fwd(r, idx, a.B, N, C, None, True)
        du = torch.randn(a.B * N, 2 * C, device="cuda", generator=g).to(dt)
        for _ in range(3):
            ops.mr_aggregate_bwd(du, idx, am, a.B, N, C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.mr_aggregate_bwd(du, idx, am, a.B, N, C)
        e1.record(); torch.cuda.synchronize()
        print(f"B={a.B} N={N:3d} C={C:3d} k={a.k} {str(dt)[6:]:8s} {e0.elapsed_time(e1) * 50:6.1f} us")
