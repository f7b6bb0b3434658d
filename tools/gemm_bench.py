#!/usr/bin/env python3
"""GEMM-family microbench on the shapes of the 't' encoder at B=256 (one view): per shape and direction, average
device time (the repetitions are captured in one hipGraph), TFLOP/s and algorithmic GB/s.
Usage: python tools/gemm_bench.py [--reps 20] [--only fwd] [--shapes MxNxKxG,...]
Tuning keys of the kernel library (--tune key=value; the defaults are the measured winners): fwd_narrow=0/1,
bwd_narrow=0/1 (64- vs 128-wide tiles), wgrad_wide=0 (never use the 8-wave 128x128 weight-gradient form),
w3_wgs / w3_min_tiles (its workgroup target / smallest layer), knn_strips=1 (strip kNN kernel)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops  # noqa: E402

SHAPES = [  # (M, Nout, K, groups, affine)
    (65536, 64, 64, 1, False), (65536, 32, 32, 4, False), (65536, 64, 128, 1, True), (65536, 256, 64, 1, False),
    (65536, 64, 256, 1, True), (32768, 128, 128, 1, False), (32768, 512, 128, 1, False), (32768, 128, 512, 1, True),
    (16384, 256, 256, 1, False), (16384, 128, 128, 4, False), (16384, 256, 512, 1, True), (16384, 1024, 256, 1, False),
    (16384, 256, 1024, 1, True), (8192, 512, 512, 1, False), (8192, 2048, 512, 1, False), (8192, 512, 2048, 1, True),
]


def run_cold(args, M, N, K, G, aff, adt, dev):
    esz = 2 if adt == torch.bfloat16 else 4
    per_set = esz * M * G * (2 * K + 2 * N)
    nb = max(2, int(600e6 // per_set) + 1)
    xs = [torch.randn(M, G * K, device=dev).to(adt) for _ in range(nb)]
    douts = [torch.randn(M, G * N, device=dev).to(adt) for _ in range(nb)]
    outs = [torch.empty(M, G * N, device=dev, dtype=adt) for _ in range(nb)]
    dins = [torch.empty(M, G * K, device=dev, dtype=adt) for _ in range(nb)]
    w = torch.randn(G * N, K, device=dev) * K ** -0.5
    if args.storage == "bf16" and not args.fp32_weights:
        ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    sc = (1 + 0.1 * torch.randn(G * K, device=dev)) if aff else None
    sh = (0.1 * torch.randn(G * K, device=dev)) if aff else None
    dw = torch.zeros(G * N, K, device=dev)
    act = ops.ACT_RELU if (aff or args.relu_in) else 0
    adds = [torch.randn(M, G * N, device=dev).to(adt) for _ in range(nb)] if args.addend else None
    runs = {
        "fwd": lambda i: ops.linear_fwd(xs[i], w, None, M, N, K, G, sc, sh, act, 0, want_stat=not (args.no_stat or args.addend),
                                        out=outs[i], addend=adds[i] if adds else None),
        "bwd_data": lambda i: ops.linear_bwd_data(douts[i], w, M, N, K, G, out=dins[i]),
        "bwd_weight": lambda i: ops.linear_bwd_weight(douts[i], xs[i], dw, M, N, K, G, sc, sh, act),
    }
    if args.blas:      # YARDSTICK ONLY (never on the product path): the vendor library's plain bf16 GEMM on the same cold operands
        wb = w.to(adt)
        dwb = torch.empty(G * N, K, device=dev, dtype=adt)
        if G != 1:
            return
        runs = {"fwd": lambda i: torch.mm(xs[i], wb.t(), out=outs[i]),
                "bwd_data": lambda i: torch.mm(douts[i], wb, out=dins[i]),
                "bwd_weight": lambda i: torch.mm(douts[i].t(), xs[i], out=dwb)}
    nbytes = {"fwd": G * (esz * M * K + 2.0 * N * K + esz * M * N), "bwd_data": G * (esz * M * N + 2.0 * N * K + esz * M * K),
              "bwd_weight": G * (esz * M * N + esz * M * K + 4.0 * N * K)}
    flops = 2.0 * M * N * K * G
    line = f"M={M:6d} N={N:5d} K={K:5d} G={G} aff={int(aff)} cold x{nb}" + (" BLAS" if args.blas else "")
    reps = max(args.reps, nb) // nb * nb
    for name in args.only.split(","):
        fn = runs[name]
        for i in range(nb):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r in range(reps):
                fn(r % nb)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        line += f" | {name} {us:6.1f}us {flops / us / 1e6:5.0f}TF {nbytes[name] / us / 1e3:5.0f}GB/s"
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="fwd,bwd_data,bwd_weight")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--aff", action="store_true", help="--shapes: the left operand carries a producer BatchNorm + ReLU (fc2-type layers)")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-stat", action="store_true", help="forward GEMM without the BatchNorm statistics epilogue")
    ap.add_argument("--fp32-weights", action="store_true", help="do not use bf16 weight shadows")
    ap.add_argument("--storage", default="bf16", help="activation storage: bf16 (needs --precision bf16) or fp32")
    ap.add_argument("--cold", action="store_true",
                    help="rotate over enough copies of the activation operands/outputs (> 600 MB) that no repetition finds "
                         "its operands in the 256 MB Infinity Cache: inside the training step every operand is cold")
    ap.add_argument("--relu-in", action="store_true", help="--cold fwd: ReLU on the operand load without an affine (eval-mode fc2)")
    ap.add_argument("--addend", action="store_true", help="--cold fwd: residual addend in the epilogue (implies no statistics)")
    ap.add_argument("--blas", action="store_true",
                    help="with --cold: time torch.mm (hipBLASLt / rocBLAS) on the same operands instead: a yardstick for the "
                         "plain product without the fused load transform / epilogues")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="tuning key of the kernel library")
    args = ap.parse_args()
    for kv in args.tune:
        k_, v_ = kv.split("=")
        ops.set_tuning(k_, int(v_))
    dev = "cuda"
    ops.set_gemm_precision(args.precision)
    adt = torch.bfloat16 if args.storage == "bf16" else torch.float32
    shapes = SHAPES
    if args.shapes:
        shapes = [tuple(int(v) for v in s.split("x")) + (args.aff,) for s in args.shapes.split(",")]
    for M, N, K, G, aff in shapes:
        if args.cold:
            run_cold(args, M, N, K, G, aff, adt, dev)
            continue
        x = torch.randn(M, G * K, device=dev).to(adt)
        w = torch.randn(G * N, K, device=dev) * K ** -0.5
        if args.storage == "bf16" and not args.fp32_weights:
            ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
        dout = torch.randn(M, G * N, device=dev).to(adt)
        sc = (1 + 0.1 * torch.randn(G * K, device=dev)) if aff else None
        sh = (0.1 * torch.randn(G * K, device=dev)) if aff else None
        dw = torch.zeros(G * N, K, device=dev)
        out = torch.empty(M, G * N, device=dev, dtype=adt)
        din = torch.empty(M, G * K, device=dev, dtype=adt)
        esz = x.element_size()
        nbytes = {"fwd": G * (esz * M * K + 4.0 * N * K + esz * M * N), "bwd_data": G * (esz * M * N + 4.0 * N * K + esz * M * K),
                  "bwd_weight": G * (esz * M * N + esz * M * K + 4.0 * N * K)}
        runs = {
            "fwd": lambda: ops.linear_fwd(x, w, None, M, N, K, G, sc, sh, ops.ACT_RELU if aff else 0, 0, want_stat=not args.no_stat,
                                          out=out),
            "bwd_data": lambda: ops.linear_bwd_data(dout, w, M, N, K, G, out=din),
            "bwd_weight": lambda: ops.linear_bwd_weight(dout, x, dw, M, N, K, G, sc, sh, ops.ACT_RELU if aff else 0),
        }
        flops = 2.0 * M * N * K * G
        line = f"M={M:6d} N={N:5d} K={K:5d} G={G}"
        for name in args.only.split(","):
            fn = runs[name]
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()               # device time per call: no host launch cost between the reps
            with torch.cuda.graph(g):
                for _ in range(args.reps):
                    fn()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
            line += f" | {name} {us:6.1f}us {flops / us / 1e6:5.0f}TF {nbytes[name] / us / 1e3:5.0f}GB/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
