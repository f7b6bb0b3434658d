#!/usr/bin/env python3
"""The reference's training loop (train.py:53-81, :117-128) on the MI355X modules, with synthetic log-mel clips instead of
its audio pipeline (audio decoding / augmentation are outside this path). Shows the drop-in: only the import lines differ.

    python tools/train_synthetic.py --steps 50                       # reference-style loop: torch.optim.Adam + clip_grad_norm_
    python tools/train_synthetic.py --steps 50 --fused --bf16        # FusedClipAdam, bf16 activation storage, two-stream views
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import functional as F_, ops  # noqa: E402
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder  # noqa: E402  (reference: encoder.graph_encoder)
from neuralsampleid_amd.optim import FusedClipAdam  # noqa: E402
from neuralsampleid_amd.simclr.ntxent import ntxent_loss  # noqa: E402     (reference: simclr.ntxent)
from neuralsampleid_amd.simclr.simclr import SimCLR  # noqa: E402          (reference: simclr.simclr)

CFG = {"arch": "grafp", "n_mels": 64, "n_frames": 128, "patch_bins": 4, "patch_frames": 8, "n_filters": 8,
       "bsz_train": 256, "tau": 0.05, "lr": 8.0e-5, "d": 128, "h": 1024, "u": 32}


def batches(n, batch, device):
    g = torch.Generator().manual_seed(0)
    for _ in range(n):
        x_i = torch.randn(batch, CFG["n_mels"], CFG["n_frames"], generator=g) * 20.0 - 40.0
        yield x_i.to(device), (x_i + 3.0 * torch.randn(x_i.shape, generator=g)).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=CFG["bsz_train"])
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--fused", action="store_true", help="FusedClipAdam instead of clip_grad_norm_ + torch.optim.Adam")
    ap.add_argument("--graph", action="store_true", help="capture the step in a hipGraph (graphs.GraphedTrainStep; needs --fused)")
    ap.add_argument("--bf16", action="store_true", help="bf16 activation storage + bf16 MFMA operands (BASELINE config 2)")
    args = ap.parse_args()
    device = torch.device("cuda")
    if args.bf16:
        ops.set_gemm_precision("bf16")
        F_.set_activation_dtype("bf16")
    torch.manual_seed(42)
    model = SimCLR(CFG, encoder=GraphEncoder(cfg=CFG, in_channels=CFG["n_filters"], k=args.k, size="t"),
                   overlap_views=args.fused).to(device)                                   # train.py:113-116
    model.train()
    if args.fused:
        optimizer = FusedClipAdam(model.parameters(), lr=CFG["lr"], max_norm=1.0)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=CFG["lr"])                    # train.py:126
    if args.graph:
        from neuralsampleid_amd.graphs import GraphedTrainStep
        data = list(batches(args.steps, args.batch, device))
        step = GraphedTrainStep(model, optimizer, CFG, *data[0])
        torch.cuda.synchronize()
        t0 = time.time()
        for idx, (x_i, x_j) in enumerate(data):
            loss = step(x_i, x_j)
            if idx % 10 == 0:
                print(f"Step [{idx}/{args.steps}]\t Loss: {loss.item():.4f}")
        torch.cuda.synchronize()
        print(f"{args.steps} graph replays in {time.time() - t0:.2f} s")
        return
    t0 = time.time()
    for idx, (x_i, x_j) in enumerate(batches(args.steps, args.batch, device)):           # train.py:53
        optimizer.zero_grad()                                                             # :58
        h_i, h_j, z_i, z_j = model(x_i, x_j)                                              # :61
        loss = ntxent_loss(z_i, z_j, CFG)                                                 # :63
        if torch.isnan(loss):                                                             # :65-68
            print(f"NaN loss at step {idx}, skipping batch")
            continue
        loss.backward()                                                                   # :70
        if not args.fused:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)              # :73
        optimizer.step()                                                                  # :75
        if idx % 10 == 0:
            print(f"Step [{idx}/{args.steps}]\t Loss: {loss.item():.4f}")                  # :77-78
    torch.cuda.synchronize()
    print(f"{args.steps} steps in {time.time() - t0:.2f} s (eager launches; bench.py replays the step as one hipGraph)")


if __name__ == "__main__":
    main()
