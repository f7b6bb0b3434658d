#!/usr/bin/env python3
"""peak-extractor patchify forward at the extraction micro-batch (2 048 clips of 64 x 128), bf16 output: launch time"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
B, H, W, pb, pf, F = 2048, 64, 128, 4, 8, 8
g = torch.Generator().manual_seed(0)
spec = torch.randn(B, H, W, generator=g).cuda()
w = (0.1 * torch.randn(F, 3, pb, pf, generator=g)).cuda(); b = (0.1 * torch.randn(F, generator=g)).cuda()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    t = timeit(lambda: ops.peak_patchify_fwd(spec, w, b, pb, pf, dt))
    print(f"patchify forward, {B} clips, output {dt}: {t:.1f} us")
