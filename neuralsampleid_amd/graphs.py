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
    valid after the replay completes; no host synchronisation is issued here).

    Construction does NOT advance the training state: the warm-up steps that precede the capture (allocator pools, lazy
    initialisation, RCCL channels) run real updates on the example batch, so the weights, both Adam moments, the step
    counter (it feeds Adam's bias correction) and every BatchNorm buffer (running statistics, num_batches_tracked) are
    snapshotted before and restored after — the first `step(x_i, x_j)` is step 1 of train.py:53-75, exactly as in the
    eager loop (tests/test_e2e_gpu.py::test_graphed_train_step_equals_eager)."""

    def __init__(self, model, optimizer, cfg: dict, x_i: torch.Tensor, x_j: torch.Tensor,
                 loss_fn: Optional[Callable] = None, reducer=None, warmup: int = 2, capture_error_mode: str = "global"):
        if not hasattr(optimizer, "flat_g"):
            raise TypeError("GraphedTrainStep needs optim.FusedClipAdam (device-side lr / step / NaN-batch skip)")
        self.model, self.opt, self.cfg, self.reducer = model, optimizer, cfg, reducer
        self.loss_fn = loss_fn or parallel.dist_ntxent_loss
        self.x_i, self.x_j = x_i.clone(), x_j.clone()
        self.loss = torch.zeros((), device=x_i.device)
        snap = self._snapshot()
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
        self._restore(snap)

    def _snapshot(self):
        opt = self.opt
        bufs = [b for b in self.model.buffers()]
        return ([t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_count, opt.grad_norm)],
                bufs, [b.clone() for b in bufs])

    def _restore(self, snap):
        from . import ops
        opt = self.opt
        with torch.no_grad():
            for dst, src in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_count, opt.grad_norm), snap[0]):
                dst.copy_(src)
            for b, v in zip(snap[1], snap[2]):
                b.copy_(v)
            opt.flat_g.zero_()
        self.loss.zero_()
        ops.bump_state_epoch()
        torch.cuda.synchronize()

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
        from . import ops
        ops.scale_f32(loss.detach().reshape(1), None, self.loss.reshape(1))

    def __call__(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        self.x_i.copy_(x_i, non_blocking=True)
        self.x_j.copy_(x_j, non_blocking=True)
        self.graph.replay()
        from . import ops
        ops.bump_state_epoch()           # the replay wrote weights and running statistics behind torch's version counters
        return self.loss
