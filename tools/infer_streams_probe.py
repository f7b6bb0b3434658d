#!/usr/bin/env python3
"""Probe: forward-only extraction with S concurrent hipGraph replays (one per HIP stream, micro-batch mb each) against the
single-stream loop. Usage (GPU box): python tools/infer_streams_probe.py [clips]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuralsampleid_amd import functional as F_, ops, fingerprint  # noqa: E402
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder  # noqa: E402
from neuralsampleid_amd.simclr.simclr import SimCLR  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
ops.set_gemm_precision("bf16")
F_.set_activation_dtype("bf16")
torch.manual_seed(42)
model = SimCLR(bench.CFG, GraphEncoder(bench.CFG, in_channels=8, k=3, size="t")).to(dev).eval()
for mb, S in [tuple(int(v) for v in a.split("x")) for a in (sys.argv[2] if len(sys.argv) > 2 else "2048x1,2048x2").split(",")]:
    pool, _ = bench.synth_clips(mb, 77, dev)
    fps = [fingerprint.GraphedFingerprinter(model, mb) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((mb, 128), device=dev) for _ in range(S)]
    for f in fps:
        f.x.copy_(pool)
    torch.cuda.synchronize()
    n = clips // (mb * S)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            for s_, f in zip(streams, fps):
                with torch.cuda.stream(s_):
                    f.graph.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"micro-batch {mb} x {S} streams: {n * mb * S / dt:9.0f} clips/s  ({1e3 * dt / n:.2f} ms per round of {mb * S} clips)", flush=True)
    del fps
