#!/usr/bin/env python3
"""knn2 with one 146-VGPR workgroup per CU against two 128-VGPR workgroups per CU (tuning key knn_pair_min), per stage shape.
Usage: python tools/knn2_pair_ab.py [--B 2048]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=2048); ap.add_argument("--k", type=int, default=3)
a = ap.parse_args()
for N, C in ((256, 64), (128, 128), (64, 256), (32, 512)):
    for dt in (torch.bfloat16, torch.float32):
        r = torch.randn(a.B * N, C, device="cuda").to(dt)
        aff = ops.BNAffine(torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1)
        res = {}
        for name, pm in (("one", 0), ("pair", 1)):
            ops.set_tuning("knn_pair_min", pm)
            for _ in range(3):
                out = ops.knn_graph(r, a.B, N, C, a.k, 1, aff)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = ops.knn_graph(r, a.B, N, C, a.k, 1, aff)
            e1.record(); torch.cuda.synchronize()
            res[name] = (e0.elapsed_time(e1) * 50, out.clone())
        same = bool((res["one"][1] == res["pair"][1]).all())
        print(f"B={a.B} N={N:3d} C={C:3d} {str(dt)[6:]:8s} one/CU {res['one'][0]:7.1f} us   pair {res['pair'][0]:7.1f} us   same ids: {same}")
ops.reset_tuning()
