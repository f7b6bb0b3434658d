#!/usr/bin/env python3
"""Soak: the train-mode forward of both views at the timed size (B = 256, bf16 storage) and the eval-mode extraction of 2 048 clips,
repeated many times on the same inputs; reports the largest deviation of z from the first pass (the head's fp32 atomics allow ~1e-7;
a dropped term or a stale tile shows as 1e-3 or more).   Usage: python tools/forward_soak.py [passes]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from synth import GRAFP_CFG
from test_e2e_gpu import build_model
from neuralsampleid_amd import fingerprint, functional as F_
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F_.set_activation_dtype("bf16")
gen = torch.Generator().manual_seed(11)
model = build_model(3).train()
x_i = (torch.randn(256, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gen) * 20 - 40).to("cuda")
x_j = x_i + 3 * torch.randn(x_i.shape, generator=gen).to("cuda")
with torch.no_grad():
    first = torch.cat(model(x_i, x_j)[2:]).clone()
    worst = torch.zeros((), device="cuda")
    for _ in range(passes):
        worst = torch.maximum(worst, (torch.cat(model(x_i, x_j)[2:]) - first).abs().max())
print(f"train-mode forward, B = 256, bf16, {passes} passes: max |z - z_first| = {float(worst):.3e}")
model.eval()
x = (torch.randn(2048, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gen) * 20 - 40).to("cuda")
first = fingerprint.extract_fingerprints(model, x, 2048).clone()
worst = torch.zeros((), device="cuda")
for _ in range(passes):
    worst = torch.maximum(worst, (fingerprint.extract_fingerprints(model, x, 2048) - first).abs().max())
print(f"eval-mode extraction, 2 048 clips, bf16, {passes} passes: max |z - z_first| = {float(worst):.3e}")
