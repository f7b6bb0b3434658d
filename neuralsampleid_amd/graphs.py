"""Whole-step hipGraph capture for training: the contrastive step is ~1 000 kernel launches of 5-40 us each, so an eager
Python loop is host-bound (~80 ms/step against ~9 ms of GPU work at batch 256). `GraphedTrainStep` captures
zero_grad -> forward (two views) -> NT-Xent -> backward -> [gradient all-reduce] -> clip + Adam once and replays it.

Everything inside the step is capture-safe by construction: the kernels never allocate or synchronise, the optimiser's
learning rate / step counter / NaN-batch skip live on the device (optim.FusedClipAdam), and the data-parallel collectives
run on the communicator's own stream (rccl.RcclComm). Inputs are copied into static buffers before every replay."""
from typing import Callable, Optional

import torch

from . import parallel


class GraphedTrainStep:
    """step = GraphedTrainStep(model, opt, cfg, x_i_example, x_j_example); loss = step(x_i, x_j)  (0-dim device tensor,
    valid after the replay completes; no host synchronisation is issued here)."""

    def __init__(self, model, optimizer, cfg: dict, x_i: torch.Tensor, x_j: torch.Tensor,
                 loss_fn: Optional[Callable] = None, reducer=None, warmup: int = 2, capture_error_mode: str = "global"):
        if not hasattr(optimizer, "flat_g"):
            raise TypeError("GraphedTrainStep needs optim.FusedClipAdam (device-side lr / step / NaN-batch skip)")
        self.model, self.opt, self.cfg, self.reducer = model, optimizer, cfg, reducer
        self.loss_fn = loss_fn or parallel.dist_ntxent_loss
        self.x_i, self.x_j = x_i.clone(), x_j.clone()
        self.loss = torch.zeros((), device=x_i.device)
        # warm-up on a side stream (allocator pools, lazy initialisation, RCCL channels), then capture
        side = torch.cuda.Stream(device=x_i.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self._step()

    def _step(self):
        self.opt.zero_grad()
        if self.reducer is not None:
            self.reducer.start_step()
            side = getattr(self.model, "_side_stream", None)
            if side is not None:
                self.reducer.streams = [torch.cuda.current_stream(), side]
        _, _, z_i, z_j = self.model(self.x_i, self.x_j)
        loss = self.loss_fn(z_i, z_j, self.cfg)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        self.loss.copy_(loss.detach())

    def __call__(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        self.x_i.copy_(x_i, non_blocking=True)
        self.x_j.copy_(x_j, non_blocking=True)
        self.graph.replay()
        return self.loss
