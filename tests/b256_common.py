"""Shared by the CPU (oracle) and GPU (HIP) tests of the TIMED configuration: B = 256, seed-42 default initialisation, the bench's
own clip pairs — fixtures tests/golden/b256_seed42_k3* (tests/golden/make_golden.py::gold_b256, from the live reference)."""
import json
import os

import numpy as np
import torch

import tapes
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


def patches_of(g, tag="s0"):
    return tapes.patches_of(g, tag)


def check_tape(tape, g, tag="s0"):
    """(hard, soft, rows) of a KnnTape that ran in patch mode: tests/tapes.py::check_patched"""
    assert tapes.n_calls(g, tag) == N_CALLS
    return tapes.check_patched(tape, g, tag)


def per_clip(t):
    t = t.detach().cpu().double()
    return torch.stack([t.sum(1), t.norm(dim=1)], 1)


def seed42_state(build):
    """the reference's default initialisation under torch.manual_seed(42) (init parity: tests/test_modules_cpu.py)"""
    torch.manual_seed(42)
    return build()
