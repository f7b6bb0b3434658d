#!/usr/bin/env python3
"""the C = 256 eval-mode FFN at the extraction micro-batch (M = 2048 clips x 64 nodes): one fused launch against the two GEMM launches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
ops.set_gemm_precision("bf16")
C = int(os.environ.get("FFN_C", "256"))
M, H = 131072 * 256 // C, 4 * C
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).to(torch.bfloat16).cuda()
w1 = (torch.randn(H, C, generator=g) * C ** -0.5).cuda(); w2 = (torch.randn(C, H, generator=g) * H ** -0.5).cuda()
b1 = (0.3 * torch.randn(H, generator=g)).cuda(); b2 = (0.3 * torch.randn(C, generator=g)).cuda()
for w in (w1, w2):
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def two():
    h, _ = ops.linear_fwd(x, w1, b1, M, H, C, act_out=ops.ACT_RELU)
    ops.linear_fwd(h, w2, b2, M, C, H, addend=x)
for a in sys.argv[1:]:
    if "=" in a: ops.set_tuning(a.split("=")[0], int(a.split("=")[1]))
t_f = timeit(lambda: ops.ffn_fused_fwd(x, w1, b1, w2, b2, M, C, H))
t_2 = float("nan") if "--fused-only" in sys.argv else timeit(two)
fl = 4.0 * M * C * H
print(f"fused {t_f:.1f} us = {fl / t_f / 1e6:.0f} TFLOP/s | two launches {t_2:.1f} us = {fl / t_2 / 1e6:.0f} TFLOP/s")
