#!/usr/bin/env python3
"""Microbench of the non-GEMM kernels on the 't' encoder's stage shapes at B=256 (one view), bf16 storage.
Usage: python tools/op_bench.py [--ops knn,mr,bn] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops  # noqa: E402

STAGES = [(256, 64), (128, 128), (64, 256), (32, 512)]


def timeit(fn, reps):
    """average device time per call: `reps` calls captured in one hipGraph (no host launch cost between them)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ops", default="knn,mr,bn")
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--storage", default="bf16")
    args = ap.parse_args()
    dev, B, k = "cuda", args.batch, args.k
    adt = torch.bfloat16 if args.storage == "bf16" else torch.float32
    which = args.ops.split(",")
    for N, C in STAGES:
        M = B * N
        r = torch.randn(M, C, device=dev).to(adt)
        sc, sh = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
        aff = ops.BNAffine(sc, sh, torch.zeros(C, device=dev), torch.ones(C, device=dev))
        line = f"N={N:4d} C={C:4d}"
        esz = r.element_size()
        if "knn" in which:
            us = timeit(lambda: ops.knn_graph(r, B, N, C, k, 1, aff), args.reps)
            line += f" | knn {us:7.1f}us {M * C * esz / us / 1e3:6.0f}GB/s"
        idx = ops.knn_graph(r, B, N, C, k, 1, aff)
        if "mr" in which:
            us = timeit(lambda: ops.mr_aggregate_fwd(r, idx, B, N, C, aff), args.reps)
            line += f" | mr_fwd {us:6.1f}us {(3 * M * C * esz + M * C + M * k * 4) / us / 1e3:6.0f}GB/s"
            u, amax = ops.mr_aggregate_fwd(r, idx, B, N, C, aff)
            du = torch.randn(M, 2 * C, device=dev).to(adt)
            us = timeit(lambda: ops.mr_aggregate_bwd(du, idx, amax, B, N, C), args.reps)
            line += f" | mr_bwd {us:6.1f}us {(3 * M * C * esz + M * C + M * k * 4) / us / 1e3:6.0f}GB/s"
        if "bn" in which:
            dout = torch.randn(M, C, device=dev).to(adt)
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            us = timeit(lambda: ops.bn_backward(dout, r, aff, ops.ACT_RELU, dg, db), args.reps)
            line += f" | bn_bwd(3 kernels) {us:6.1f}us {5 * M * C * esz / us / 1e3:6.0f}GB/s"
            stat = torch.randn(2, ops.row_tiles(M), C, device=dev).abs()
            g, bta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), device=dev, dtype=torch.int64)
            us = timeit(lambda: ops.bn_finalize(stat, M, g, bta, rm, rv, nbt), args.reps)
            line += f" | bn_finalize {us:5.1f}us"
            us = timeit(lambda: ops.bn_apply(r, aff, ops.ACT_NONE, residual=dout), args.reps)
            line += f" | bn_apply+res {us:5.1f}us {3 * M * C * esz / us / 1e3:6.0f}GB/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
