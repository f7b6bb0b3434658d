"""clip_grad_norm_(max_norm) + Adam as two HIP kernels over ONE flat parameter buffer (train.py:73-75, :126).

Parameters, gradients and both Adam moments live in four flat fp32 buffers; every nn.Parameter (and its .grad)
becomes a 16-byte-aligned view, so the whole optimiser is 2 launches, the gradient all-reduce is one contiguous
buffer (parallel.py), and nothing here synchronises with the host (hipGraph-capturable; lr lives on the device)."""
from typing import Iterable, List

import torch

from . import ops


class FusedClipAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 8e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 1.0, direct_grads: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedClipAdam needs parameters on the MI355X (cuda) device")
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 7) // 8 * 8              # every fp32 view AND its bf16 shadow view 16-byte aligned
        self.numel = total
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.flat_p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + p.numel()].view_as(p)
        # bf16 shadow of every parameter for the bf16-storage GEMMs (ops.WeightShadows): one flat buffer, refreshed by
        # one launch at the start of each step (zero_grad) — 18.4 M parameters = 110 MB of traffic, ~0.2 % of a step
        self.flat_p16 = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        ops.SHADOWS.purge()
        for p, off in zip(self.params, self.offsets):
            if p.numel() % 8 == 0:
                ops.SHADOWS.register(p.data, self.flat_p16[off:off + p.numel()], owner=self.flat_p)
        # Downsample conv weights (Cout, Cin, 3, 3): packed forms + shared packed gradient prepared once per step (ops.DsPrep)
        for p in self.params:            # (register purges the entries of weights that are gone or were re-flattened)
            if p.dim() == 4 and tuple(p.shape[2:]) == (3, 3):
                ops.DS_PREP.register(p)
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, max_norm if max_norm else 0.0], device=dev)
        self.step_count = torch.zeros((), dtype=torch.int64, device=dev)
        self.grad_norm = torch.zeros(1, device=dev)       # pre-clip global norm of the last step
        if direct_grads:      # block backward kernels accumulate straight into the flat gradient buffer
            from . import functional
            functional.DIRECT_GRADS = True

    def zero_grad(self, set_to_none: bool = False):
        from . import functional
        functional.DEFERRED.reset()          # (a backward pass that raised may have left recorded weight-gradient problems behind)
        functional.join_side_streams()
        ops.fill_zero(self.flat_g)
        if functional.ACT_DTYPE == torch.bfloat16:
            self.sync_shadow()
            ops.DS_PREP.refresh()        # (the step's two views run behind this point on both streams)

    def sync_shadow(self):
        """refresh the bf16 weight shadows from the fp32 master parameters"""
        n8 = self.numel                      # a multiple of 8 by construction
        ops.f32_to_bf16(self.flat_p[:n8], self.flat_p16)
        for p in self.params:                # every registered view now holds the conversion of the current weights
            ops.SHADOWS.mark_fresh(p)        # (the Parameter, not p.data: a .data alias carries its own version counter)

    def set_lr(self, lr: float):
        self.hyper[0:1].fill_(lr)

    @property
    def lr(self) -> float:
        return float(self.hyper[0])

    def step(self):
        from . import functional
        functional.join_side_streams()               # a view's backward may still be running on its side stream
        partial = ops.sumsq_partial(self.flat_g)
        ops.bump_state_epoch(stats=False)            # the update kernel writes the weights behind torch's version counters
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.hyper, self.step_count, partial,
                      self.grad_norm)
