"""Shared by the CPU (oracle) and GPU (HIP) tests of the TIMED configuration: B = 256, seed-42 default initialisation, the bench's
own clip pairs — fixtures tests/golden/b256_seed42_k3* (tests/golden/make_golden.py::gold_b256, from the live reference)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN
from synth import GRAFP_CFG

B = 256
N_CALLS = 24          # 12 blocks x 2 views, view i first


def bench_clips(batch=B, seed=1000):
    """bench.py's synth_clips on the CPU (same generator calls, same values)"""
    gi = torch.Generator().manual_seed(seed)
    gj = torch.Generator().manual_seed(seed + 1)
    x_i = torch.randn(batch, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gi) * 20.0 - 40.0
    x_j = x_i + 3.0 * torch.randn(batch, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gj)
    return x_i, x_j


def checksums():
    with open(os.path.join(GOLDEN, "b256_seed42_k3_checksums.json")) as f:
        return json.load(f)


def chaos():
    with open(os.path.join(GOLDEN, "b256_seed42_k3_chaos.json")) as f:
        return json.load(f)


def tape_of(g, tag="s0"):
    return [g.t(f"knn.{tag}.{c}").to(torch.int64) for c in range(N_CALLS)]


def check_tape(recorded, g, tag="s0"):
    """own neighbour SETS against the reference's: (hard mismatches, mismatches on recorded near-tie rows, rows)"""
    hard = soft = rows = 0
    for c, own in enumerate(recorded):
        ref = g[f"knn.{tag}.{c}"].astype(np.int64)
        own = own.detach().cpu().numpy().astype(np.int64)
        near = np.unpackbits(g[f"near.{tag}.{c}"])[: ref.shape[0] * ref.shape[1]].reshape(ref.shape[:2]).astype(bool)
        diff = (np.sort(own, -1) != np.sort(ref, -1)).any(-1)
        hard += int((diff & ~near).sum())
        soft += int((diff & near).sum())
        rows += diff.size
    return hard, soft, rows


def per_clip(t):
    t = t.detach().cpu().double()
    return torch.stack([t.sum(1), t.norm(dim=1)], 1)


def seed42_state(build):
    """the reference's default initialisation under torch.manual_seed(42) (init parity: tests/test_modules_cpu.py)"""
    torch.manual_seed(42)
    return build()
