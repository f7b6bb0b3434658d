#!/usr/bin/env python3
"""the launches of one eval-mode C = 256 block at the extraction micro-batch (2 048 clips x 64 nodes), each timed alone (warm):
grapher fc1 (256 -> 256), fc2 + shortcut (512 -> 256), the fused FFN -- what a wider fusion could save"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
ops.set_gemm_precision("bf16")
M, C = 131072, 256
g = torch.Generator().manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale)
x0 = rnd(M, C).to(torch.bfloat16).cuda(); r2 = rnd(M, 2 * C).to(torch.bfloat16).cuda()
wfc1, bfc1 = rnd(C, C, scale=C ** -0.5).cuda(), rnd(C).cuda()
wfc2, bfc2 = rnd(C, 2 * C, scale=(2 * C) ** -0.5).cuda(), rnd(C).cuda()
w1, b1 = rnd(4 * C, C, scale=C ** -0.5).cuda(), rnd(4 * C).cuda()
w2, b2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5).cuda(), rnd(C).cuda()
for w in (wfc1, wfc2, w1, w2):
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t1 = timeit(lambda: ops.linear_fwd(x0, wfc1, bfc1, M, C, C))
t2 = timeit(lambda: ops.linear_fwd(r2, wfc2, bfc2, M, C, 2 * C, addend=x0))
t3 = timeit(lambda: ops.ffn_fused_fwd(x0, w1, b1, w2, b2, M, C, 4 * C))
N, k = 64, 3
B = M // N
y = rnd(M, C).to(torch.bfloat16).cuda()
idx = torch.randint(0, N, (B, N, k), generator=g).to(torch.int32).cuda()
wg, bg = rnd(2 * C, C // 2, scale=(C // 2) ** -0.5).cuda(), rnd(2 * C).cuda()
ops.SHADOWS.register(wg, ops.f32_to_bf16(wg), owner=wg)
t4 = timeit(lambda: ops.mrconv_fused_fwd(y, idx, B, N, C, wg, bg))
t5 = timeit(lambda: ops.block_tail_fused_fwd(x0, r2, wfc2, bfc2, w1, b1, w2, b2, M, C, 4 * C))
t6 = timeit(lambda: ops.block_gr_fused_fwd(x0, y, idx, B, N, wg, bg, wfc2, bfc2, w1, b1, w2, b2, M, C, 4 * C))
print(f"mrconv {t4:.1f} us | tail+FFN (PRE) {t5:.1f} us | graph conv+tail+FFN (GR) {t6:.1f} us")
print(f"fc1 256->256 {t1:.1f} us | fc2 512->256 + shortcut {t2:.1f} us | fused FFN {t3:.1f} us")
