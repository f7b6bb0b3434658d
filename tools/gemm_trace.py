#!/usr/bin/env python3
"""Per-workgroup timeline of one GEMM launch (nsid_debug_gemm_trace): when each workgroup started, how long its main loop
and epilogue took, on which XCD / CU it ran and how many ran on a CU at once. Answers "where does a 20 us GEMM spend its
time" without a profiler.  Usage: python tools/gemm_trace.py [--shape MxNxK] [--dir fwd|bwd_data|bwd_weight]"""
import argparse, os, sys, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
from neuralsampleid_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16384x1024x256")
ap.add_argument("--dir", default="fwd")
ap.add_argument("--no-stat", action="store_true")
ap.add_argument("--aff", action="store_true", help="fwd / bwd_weight: the input carries a BatchNorm affine + ReLU applied on load")
ap.add_argument("--bn", type=int, default=-1, help="bwd_data: also emit the BatchNorm-backward sums (activation code 0/1/2)")
ap.add_argument("--addend", action="store_true", help="bwd_data: add a residual gradient in the epilogue")
ap.add_argument("--cold", action="store_true", help="flush caches with a 1 GiB write before the traced launch")
args = ap.parse_args()
M, N, K = (int(v) for v in args.shape.split("x"))
ops.set_gemm_precision("bf16")
dev = "cuda"
x = torch.randn(M, K, device=dev).bfloat16()
w = torch.randn(N, K, device=dev) * K ** -0.5
ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
dout = torch.randn(M, N, device=dev).bfloat16()
dw = torch.zeros(N, K, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
din = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
addend = torch.randn(M, K, device=dev).bfloat16() if args.addend else None
bn = None
if args.bn >= 0:
    r_ = torch.randn(M, K, device=dev).bfloat16()
    aff = ops.BNAffine(torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.zeros(K, device=dev),
                       torch.ones(K, device=dev))
    bn = (r_, aff, args.bn)
sc = (1 + 0.1 * torch.randn(K, device=dev)) if args.aff else None
sh = (0.1 * torch.randn(K, device=dev)) if args.aff else None
run = {"fwd": lambda: ops.linear_fwd(x, w, None, M, N, K, 1, sc, sh, ops.ACT_RELU if args.aff else 0, 0,
                                     want_stat=not args.no_stat, out=out),
       "bwd_data": lambda: ops.linear_bwd_data(dout, w, M, N, K, 1, addend, out=din, bn=bn),
       "bwd_weight": lambda: ops.linear_bwd_weight(dout, x, dw, M, N, K, 1, sc, sh, ops.ACT_RELU if args.aff else 0)}[args.dir]
for _ in range(5):
    run()
torch.cuda.synchronize()
buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
assert lib.nsid_debug_gemm_trace(buf.data_ptr()) == 0
if args.cold:
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    junk.fill_(1.0)
    torch.cuda.synchronize()
run()
torch.cuda.synchronize()
lib.nsid_debug_gemm_trace(None)
t = buf.cpu().numpy().reshape(-1, 4)
t = t[t[:, 0] != 0]
n = len(t)
t0, tl, te, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
base = t0.min()
us = lambda v: (v - base) / 100.0
xcc = (hw >> 32) & 0xF
pro = hw >> 40                      # gemm256.hip only: ticks from start to "first sub-tile landed"
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
where = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"{args.dir} {M}x{N}x{K}: {n} workgroups on {len(set(where))} CUs; span {us(te).max():.2f} us (first start -> last end)")
print(f"  start times: median {np.median(us(t0)):.2f}  p90 {np.percentile(us(t0), 90):.2f}  max {us(t0).max():.2f} us")
print(f"  workgroup duration: median {np.median(te - t0) / 100:.2f}  p10 {np.percentile(te - t0, 10) / 100:.2f}  p90 {np.percentile(te - t0, 90) / 100:.2f} us;"
      f"  main loop median {np.median(tl - t0) / 100:.2f}, epilogue median {np.median(te - tl) / 100:.2f} us")
if pro.max() > 0:
    print(f"  prologue (start -> first sub-tile in LDS): median {np.median(pro) / 100:.2f}  p90 {np.percentile(pro, 90) / 100:.2f} us")
per = collections.defaultdict(list)
for i in range(n):
    per[where[i]].append((t0[i], te[i]))
conc, busy, cnt = [], [], []
for k, iv in per.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    c = m = 0
    last, b = None, 0
    for tt, d in ev:
        if c > 0:
            b += tt - last
        c += d
        m = max(m, c)
        last = tt
    conc.append(m); busy.append(b / 100.0); cnt.append(len(iv))
print(f"  per CU: workgroups {min(cnt)}..{max(cnt)} (mean {np.mean(cnt):.2f}); max resident at once {min(conc)}..{max(conc)};"
      f" time with >= 1 resident: mean {np.mean(busy):.2f} us")
# rounds: start-time histogram
h, edges = np.histogram(us(t0), bins=12)
print("  start histogram (us):", " ".join(f"{edges[i]:.1f}:{h[i]}" for i in range(len(h))))
# duration by XCD and by position in the grid (stragglers set the kernel time)
dur = (te - t0) / 100.0
print("  duration by XCD (median us):", " ".join(f"{x}:{np.median(dur[xcc == x]):.1f}" for x in sorted(set(xcc))))
q = np.array_split(np.arange(n), 8)
print("  duration by launch-order octile (median us):", " ".join(f"{np.median(dur[i]):.1f}" for i in q))
print("  end-time percentiles (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(us(te), [50, 90, 99, 100])))
