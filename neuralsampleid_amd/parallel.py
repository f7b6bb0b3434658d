"""Data-parallel contrastive step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference scales with single-process nn.DataParallel (train.py:117-120): the batch is scattered, z is gathered to
GPU 0, NT-Xent runs over the GLOBAL batch, gradients are reduced to GPU 0 and BatchNorm statistics stay per replica.
The MI355X-native equivalent keeps those semantics with two collectives and no parameter broadcast:

  1. all-gather of the L2-normalised embeddings z_i, z_j (B_local x d each: 128 KB per rank at 256 x 128 fp32);
     every rank then evaluates the loss rows it owns against all 2*B_global columns (csrc/ntxent.hip takes the row
     range), which yields d(global mean loss)/dz for its own pairs with no second exchange;
  2. SUM all-reduce of the flat gradient buffer (73.5 MB fp32) — the per-rank gradients are already gradients of the
     global mean loss through that rank's clips, so they add (no 1/world averaging);
  BatchNorm statistics are per rank (no SyncBN), as under DataParallel.

The collective plumbing is device-agnostic (`gloo` on CPU in tests); the loss rows come from an injected kernel."""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous pairs [p0, p0+n) owned by `rank`; global_batch must divide evenly (train.py uses drop_last)"""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    n = global_batch // world
    return rank * n, n


def gather_embeddings(z: torch.Tensor, group=None) -> torch.Tensor:
    """(B_local, d) -> (B_global, d), rank-major so that global pair p = rank*B_local + local index"""
    world = dist.get_world_size(group)
    if world == 1:
        return z
    out = torch.empty((world * z.shape[0], z.shape[1]), device=z.device, dtype=z.dtype)
    dist.all_gather_into_tensor(out, z.contiguous(), group=group)
    return out


class _DistNtxent(torch.autograd.Function):
    """loss = global-mean NT-Xent; backward hands back d loss / d z_local from the sharded kernel (no collective)."""

    @staticmethod
    def forward(ctx, z_i, z_j, tau, rows_fn, group):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        zi_all = gather_embeddings(z_i.detach(), group)
        zj_all = gather_embeddings(z_j.detach(), group)
        p0, n = shard_range(zi_all.shape[0], rank, world)
        part, dzi, dzj = rows_fn(zi_all, zj_all, tau, p0, n)        # part = sum of owned rows / (2*B_global)
        loss = part.reshape(()).clone()
        if world > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)   # every rank reports the global loss
        ctx.save_for_backward(dzi, dzj)
        return loss

    @staticmethod
    def backward(ctx, g):
        dzi, dzj = ctx.saved_tensors
        return dzi * g, dzj * g, None, None, None


def _hip_rows(zi_all, zj_all, tau, p0, n):
    from . import ops
    return ops.ntxent_fwd_bwd(zi_all, zj_all, tau, p0, n, want_grad=True)


def dist_ntxent_loss(z_i: torch.Tensor, z_j: torch.Tensor, cfg: dict, group=None,
                     rows_fn: Optional[Callable] = None) -> torch.Tensor:
    """Drop-in for simclr.ntxent.ntxent_loss under data parallelism: negatives come from the GLOBAL batch."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        from .simclr.ntxent import ntxent_loss
        if rows_fn is None:
            return ntxent_loss(z_i, z_j, cfg)
    return _DistNtxent.apply(z_i, z_j, float(cfg["tau"]), rows_fn or _hip_rows, group)


def allreduce_gradients(flat_grad: torch.Tensor, group=None, bucket_bytes: int = 0, async_op: bool = False):
    """SUM the flat gradient buffer over ranks. bucket_bytes > 0 splits it into contiguous buckets (each its own
    collective, so early buckets can overlap the rest of backward when launched from a side stream)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    if bucket_bytes <= 0:
        w = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return [w] if async_op else []
    n = max(1, bucket_bytes // flat_grad.element_size())
    works = []
    for s in range(0, flat_grad.numel(), n):
        w = dist.all_reduce(flat_grad[s:s + n], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT. Returns (rank, local, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
