"""Forward-only fingerprint extraction (the reference's generate.py path, :31-49, :68-100 — BASELINE config 5).

generate.py builds SimCLR(GraphEncoder(k=3)), feeds each audio's (S, 64, 128) log-mel segments through `model(x, x)` in
splits of 128 and keeps z_i. Differences here, all deliberate and documented in DESIGN.md:
  * eval-mode BatchNorm (running statistics) — generate.py never calls model.eval() (test_fp.py:315 does);
  * one view, not two: `model(x, x)` computes the same embedding twice;
  * every split's embeddings are kept (generate.py:44-48 appends only the last one);
  * larger micro-batches (default 1024) and contiguous sharding over ranks with no collective."""
from typing import Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous clip range [lo, hi) of `rank`; the first n % world ranks take one extra clip"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@torch.no_grad()
def extract_fingerprints(model, specs: torch.Tensor, batch: int = 1024, out: torch.Tensor = None) -> torch.Tensor:
    """specs (S, n_mels, n_frames) fp32 on the GPU -> (S, d) L2-normalised fingerprints (fp32).
    The model is run in eval mode (restored afterwards)."""
    was_training = model.training
    model.eval()
    from . import functional, ops
    if functional.ACT_DTYPE == torch.bfloat16:
        ops.register_weight_shadows(model)          # bf16 copies of the weights, converted once (version-checked)
    try:
        S = specs.shape[0]
        d = model.projector[-1].out_features
        if out is None:
            out = torch.empty((S, d), device=specs.device, dtype=torch.float32)
        for lo in range(0, S, batch):
            hi = min(S, lo + batch)
            _, z = model._embed(specs[lo:hi].contiguous())
            out[lo:hi].copy_(z)
        return out
    finally:
        model.train(was_training)


@torch.no_grad()
def fingerprints_from_waveform(model, front, wave: torch.Tensor, batch: int = 1024) -> torch.Tensor:
    """mono fp32 waveform on the GPU -> (S, d) fingerprints: log-mel front end (frontend.LogMelFrontEnd, the reference's
    MelSpectrogram + AmplitudeToDB + 87.5 %-overlap unfold) fused ahead of the encoder — SURVEY.md §8f-4."""
    segs = front(wave)
    if segs.shape[0] == 0:
        return torch.empty((0, model.projector[-1].out_features), device=wave.device)
    return extract_fingerprints(model, segs, batch)
