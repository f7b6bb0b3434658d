#!/usr/bin/env python3
"""Dump nsid_knn_graph results for a set of seeded inputs (A/B of two kernel libraries: run once per NSID_LIB, compare the files).
Usage: NSID_LIB=... python tools/knn_dump.py out.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
out = {}
for seed, (B, N, C, k, d, dt) in enumerate([(64, 256, 64, 18, 1, torch.float32), (64, 256, 64, 18, 1, torch.bfloat16),
                                            (32, 256, 64, 18, 2, torch.float32), (32, 256, 64, 9, 3, torch.float32),
                                            (16, 256, 64, 30, 2, torch.bfloat16)]):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(B * N, C, generator=g).to("cuda").to(dt)
    if seed % 2 == 0:       # low-rank features: many near ties
        basis = torch.randn(4, C, generator=g).to("cuda")
        r = (torch.randn(B * N, 4, generator=g).to("cuda") @ basis).to(dt)
    aff = ops.BNAffine(torch.rand(C, generator=g).to("cuda") + 0.5, torch.randn(C, generator=g).to("cuda") * 0.1)
    idx = ops.knn_graph(r, B, N, C, k, d, aff)
    out[f"case{seed}"] = idx.cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: v.shape for k, v in out.items()})
