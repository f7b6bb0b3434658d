#!/usr/bin/env python3
"""max-relative aggregation at the deep configuration's shapes (batch 256, k = 18, bf16, BatchNorm affine on load): the integer-key
search against the scalar search (tuning key mr_key_min_k), forward and backward launch times"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, k = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 18
for a in sys.argv[2:]:
    key, _, val = a.partition("=")
    ops.set_tuning(key, int(val))
    print("tuning", key, val)
g = torch.Generator().manual_seed(0)
for N, C in ((256, 64), (128, 128), (64, 256), (32, 512)):
    r = torch.randn(B * N, C, generator=g).to(torch.bfloat16).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g).to(torch.int32).cuda()
    aff = ops.BNAffine((1 + 0.3 * torch.randn(C, generator=g)).cuda(), (0.5 * torch.randn(C, generator=g)).cuda())
    du = torch.randn(B * N, 2 * C, generator=g).to(torch.bfloat16).cuda()
    t_key = timeit(lambda: ops.mr_aggregate_fwd(r, idx, B, N, C, aff))
    _, amax = ops.mr_aggregate_fwd(r, idx, B, N, C, aff)
    t_bwd = timeit(lambda: ops.mr_aggregate_bwd(du, idx, amax, B, N, C))
    keep = ops.get_tuning("mr_key_min_k")
    ops.set_tuning("mr_key_min_k", 0)
    t_old = timeit(lambda: ops.mr_aggregate_fwd(r, idx, B, N, C, aff))
    ops.set_tuning("mr_key_min_k", keep)
    print(f"N={N:3d} C={C:3d} k={k}: forward {t_key:.1f} us (scalar search {t_old:.1f} us) | backward {t_bwd:.1f} us")
