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

Two transports behind the same three calls (all-gather, all-reduce, bucketed async all-reduce):
  * `rccl.RcclComm` (GPU, the product path): direct RCCL on one dedicated HIP stream — capturable in the step's hipGraph,
    no helper thread (see rccl.py for why ProcessGroupNCCL is not used on the data path). `init_from_env("rccl")`
    creates it (torch.distributed/gloo only carries the 128-byte ncclUniqueId and the host barrier) and installs it as
    the module default `COMM`.
  * torch.distributed collectives on the default group (`gloo` on CPU in tests, where the loss rows come from an
    injected oracle kernel; also "nccl" if a caller insists)."""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous pairs [p0, p0+n) owned by `rank`; global_batch must divide evenly (train.py uses drop_last)"""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    n = global_batch // world
    return rank * n, n


COMM = None      # default rccl.RcclComm of this process (set by init_from_env("rccl") / set_default_comm)


def set_default_comm(comm) -> None:
    global COMM
    COMM = comm


def _comm_for(group):
    """the RCCL communicator to use, or None -> torch.distributed on `group`"""
    return COMM if group is None else None


def _rank_world(group=None) -> Tuple[int, int]:
    c = _comm_for(group)
    if c is not None:
        return c.rank, c.world
    if dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def gather_embeddings(z: torch.Tensor, group=None) -> torch.Tensor:
    """(B_local, d) -> (B_global, d), rank-major so that global pair p = rank*B_local + local index"""
    if not _distributed(group):
        return z
    c = _comm_for(group)
    if c is not None:
        return c.all_gather(z)
    world = dist.get_world_size(group)
    out = torch.empty((world * z.shape[0], z.shape[1]), device=z.device, dtype=z.dtype)
    dist.all_gather_into_tensor(out, z.contiguous(), group=group)
    return out


# The all-reduce of the scalar loss only serves reporting (every rank logs the global loss). With ASYNC_LOSS_REDUCE it is
# left running on the communicator's stream instead of being joined before backward; the caller must then join that stream
# before reading the loss on the host — GradReducer.finish() and allreduce_gradients(async_op=False) both do.
ASYNC_LOSS_REDUCE = False


def gather_pair(z_i: torch.Tensor, z_j: torch.Tensor, group=None):
    """both views' embeddings of every rank, rank-major: ONE all-gather of the stacked (2, B, d) block on the RCCL path
    (each collective is a latency on the critical path between forward and backward)"""
    c = _comm_for(group)
    if c is None or not _distributed(group):
        return gather_embeddings(z_i, group), gather_embeddings(z_j, group)
    B, d = z_i.shape
    allz = c.all_gather(torch.stack((z_i, z_j)))                       # (world*2, B, d)
    return _split_views(allz, c.world)


def _split_views(allz: torch.Tensor, world: int):
    """(world*2, B, d) rank-major stacked views -> (world*B, d) per view, global pair p = rank*B + b (as gather_embeddings)"""
    _, B, d = allz.shape
    allz = allz.view(world, 2, B, d).transpose(0, 1).reshape(2, world * B, d)
    return allz[0], allz[1]


class _DistNtxent(torch.autograd.Function):
    """loss = global-mean NT-Xent; backward hands back d loss / d z_local from the sharded kernel (no collective)."""

    @staticmethod
    def forward(ctx, z_i, z_j, tau, rows_fn, group):
        rank, world = _rank_world(group)
        zi_all, zj_all = gather_pair(z_i.detach(), z_j.detach(), group)
        p0, n = shard_range(zi_all.shape[0], rank, world)
        part, dzi, dzj = rows_fn(zi_all, zj_all, tau, p0, n)        # part = sum of owned rows / (2*B_global)
        from . import ops
        loss = (ops.scale_f32(part) if part.is_cuda else part.clone()).reshape(())
        if _distributed(group):                                        # every rank reports the global loss
            c = _comm_for(group)
            if c is not None and ASYNC_LOSS_REDUCE:
                c.all_reduce_async_(loss)
            elif c is not None:
                c.all_reduce_(loss)
            else:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
        ctx.save_for_backward(dzi, dzj)
        return loss

    @staticmethod
    def backward(ctx, g):
        dzi, dzj = ctx.saved_tensors
        if dzi.is_cuda:
            from . import ops
            return ops.scale_f32(dzi, g), ops.scale_f32(dzj, g), None, None, None
        return dzi * g, dzj * g, None, None, None


def _hip_rows(zi_all, zj_all, tau, p0, n):
    from . import ops
    return ops.ntxent_fwd_bwd(zi_all, zj_all, tau, p0, n, want_grad=True)


def dist_ntxent_loss(z_i: torch.Tensor, z_j: torch.Tensor, cfg: dict, group=None,
                     rows_fn: Optional[Callable] = None) -> torch.Tensor:
    """Drop-in for simclr.ntxent.ntxent_loss under data parallelism: negatives come from the GLOBAL batch."""
    if not _distributed(group):
        from .simclr.ntxent import ntxent_loss
        if rows_fn is None:
            return ntxent_loss(z_i, z_j, cfg)
    return _DistNtxent.apply(z_i, z_j, float(cfg["tau"]), rows_fn or _hip_rows, group)


def allreduce_gradients(flat_grad: torch.Tensor, group=None, bucket_bytes: int = 0, async_op: bool = False):
    """SUM the flat gradient buffer over ranks. bucket_bytes > 0 splits it into contiguous buckets (each its own
    collective, so early buckets can overlap the rest of backward when launched from a side stream)."""
    if not _distributed(group):
        return []
    c = _comm_for(group)
    if c is not None:                      # RCCL: stream-ordered on the comm stream; "works" are joined by wait_async()
        n = flat_grad.numel() if bucket_bytes <= 0 else max(1, bucket_bytes // flat_grad.element_size())
        for s in range(0, flat_grad.numel(), n):
            c.all_reduce_async_(flat_grad[s:s + n])
        if not async_op:
            c.wait_async()
        return [c] if async_op else []
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


def _distributed(group=None) -> bool:
    """collectives are issued when a process group exists and has >1 rank (NSID_FORCE_COLLECTIVES=1: also with one
    rank, so that the collective code path — including hipGraph capture of it — can be exercised on a single GPU)"""
    import os
    force = os.environ.get("NSID_FORCE_COLLECTIVES", "0") == "1"
    c = _comm_for(group)
    if c is not None:
        return c.world > 1 or force
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or force


class GradReducer:
    """Bucketed SUM all-reduce of the flat gradient buffer, overlapped with backward.

    Parameters sit in the flat buffer in registration (= forward) order and backward completes them last-to-first, so
    buckets are cut from the END of the buffer. Every parameter receives `uses_per_step` contributions per step (two:
    the encoder runs once per view); when the last contribution of a bucket has been ENQUEUED, its all-reduce is
    launched with async_op=True — the collective's stream waits for the compute stream at that point and then runs
    concurrently with the remaining backward kernels (xGMI links are otherwise idle during backward). `finish()`
    reduces whatever did not fire and makes the compute stream wait for all of it before the optimiser step."""

    def __init__(self, params, flat_grad: torch.Tensor, offsets, group=None, bucket_bytes: int = 16 << 20,
                 uses_per_step: int = 2):
        self.flat, self.group, self.uses = flat_grad, group, uses_per_step
        self.bucket_of, self.bounds, self.need = {}, [], []
        esz = flat_grad.element_size()
        end = flat_grad.numel()
        cur_start, count = end, 0
        for p, off in zip(reversed(list(params)), reversed(list(offsets))):
            cur_start = off
            count += 1
            self.bucket_of[id(p)] = len(self.bounds)
            if (end - cur_start) * esz >= bucket_bytes:
                self.bounds.append((cur_start, end))
                self.need.append(count * uses_per_step)
                end, count = cur_start, 0
        if count:
            self.bounds.append((0, end))
            self.need.append(count * uses_per_step)
        self.remaining, self.works, self.fired = [], [], []
        self.streams = []                            # extra compute streams (two-stream views), see _fire

    def start_step(self):
        self.remaining = list(self.need)
        self.works, self.fired = [], []

    def block_done(self, params):
        for p in params:
            b = self.bucket_of.get(id(p))
            if b is None:
                continue
            self.remaining[b] -= 1
            if self.remaining[b] == 0:
                self._fire(b)

    def _fire(self, b: int):
        self.fired.append(b)
        if _distributed(self.group):
            s, e = self.bounds[b]
            cur = torch.cuda.current_stream() if self.flat.is_cuda else None
            c = _comm_for(self.group)
            from . import functional
            others = [st for st in list(self.streams) + list(functional.SIDE_STREAMS) if st != cur]
            if c is not None:                        # the comm stream forks from every stream that carried contributions
                c.all_reduce_async_(self.flat[s:e], producers=[cur] + others)
                return
            for st in others:                        # contributions may have been enqueued on another view's stream
                if cur is not None and st != cur:
                    cur.wait_stream(st)
            self.works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        from . import functional
        functional.join_side_streams()
        for b, r in enumerate(self.remaining):
            if r > 0:                      # a parameter that got fewer contributions than expected this step
                self.remaining[b] = 0
                self._fire(b)
        for w in self.works:
            w.wait()
        self.works = []
        c = _comm_for(self.group)
        if c is not None and _distributed(self.group):
            c.wait_async()                 # the compute stream waits for the comm stream (no host sync)

    def install(self):
        from . import functional
        functional.GRAD_READY_HOOK = self.block_done
        if _distributed(self.group):
            # the deferred weight-gradient phase (functional.DeferredWgrads) goes out in three pieces, late layers first, so that the
            # all-reduce of a piece's buckets overlaps the launches of the next one
            functional.DEFER_CHUNKS = 3
        return self


# Host-side deadlines. The direct RCCL path has no ProcessGroupNCCL watchdog (by design, rccl.py), so the host waits are
# bounded instead: the bootstrap group gets an explicit timeout, and `sync_with_deadline` replaces a bare
# torch.cuda.synchronize() wherever a peer's death would otherwise block this rank inside ncclAllReduce / a graph replay
# for ever. On expiry (or an asynchronous RCCL error) the communicator is aborted and the process exits non-zero; the
# launcher (torch.distributed.run, or bench.py's own spawner) then ends the remaining ranks. Never re-exec here.
HOST_TIMEOUT_S = float(__import__("os").environ.get("NSID_HOST_TIMEOUT_S", "300"))


class CollectiveTimeout(RuntimeError):
    pass


def sync_with_deadline(seconds: Optional[float] = None, what: str = "device work", fatal: bool = True) -> None:
    """torch.cuda.synchronize() with a host deadline: an event recorded on every known stream is polled, the communicator's
    asynchronous error state is checked meanwhile. fatal: abort the communicator and os._exit(1).
    Polling: 50 us sleeps for the first 5 s (a bench's timed region ends here: a 1 ms poll interval would add up to 0.6 % to a
    160 ms measurement), 1 ms afterwards; the RCCL error query runs once per millisecond, not once per poll."""
    import os
    import sys
    import time
    seconds = HOST_TIMEOUT_S if seconds is None else seconds
    from . import functional
    streams = [torch.cuda.current_stream()] + [s for s in functional.SIDE_STREAMS]
    if COMM is not None:
        streams.append(COMM.stream)
    evs = []
    for st in streams:
        e = torch.cuda.Event()
        e.record(st)
        evs.append(e)
    t0 = time.monotonic()
    err = None
    last_check = t0
    while True:
        if all(e.query() for e in evs):
            return
        now = time.monotonic()
        if now - t0 < 5.0 and now - last_check < 1e-3:
            time.sleep(5e-5)
            continue
        last_check = now
        if COMM is not None and COMM._h is not None:
            try:
                code = COMM.async_error()
            except RuntimeError as ex:          # the query itself failed
                code, err = -1, str(ex)
            if code != 0:
                err = err or f"RCCL asynchronous error {code}"
                break
        if time.monotonic() - t0 > seconds:
            err = f"{what} did not finish within {seconds:.0f} s"
            break
        time.sleep(5e-5 if now - t0 < 5.0 else 1e-3)
    if not fatal:
        raise CollectiveTimeout(err)
    print(f"[nsid] rank {_rank_world()[0]}: {err}; aborting the communicator and exiting", file=sys.stderr, flush=True)
    if COMM is not None:
        try:
            COMM.abort()
        except Exception:
            pass
    os._exit(1)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT. Returns (rank, local, world).

    backend "rccl" (the GPU product path): torch.distributed over gloo for bootstrap + host barriers, and a direct
    rccl.RcclComm (installed as the default COMM) for every data-path collective. "gloo"/"nccl": torch.distributed only."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("NSID_FORCE_COLLECTIVES", "0") == "1"
    if backend is None:
        backend = "rccl" if torch.cuda.is_available() else "gloo"
    if backend in ("rccl", "nccl"):
        torch.cuda.set_device(local)
    from .rccl import _stdout_to_stderr         # gloo announces "[Gloo] Rank r is connected to ..." on stdout: a program that
    if world > 1 and not dist.is_initialized():  # prints one machine-readable line there (bench.py) must not carry it
        import datetime
        to = datetime.timedelta(seconds=HOST_TIMEOUT_S)     # a rank that never arrives fails the others, it does not hang them
        with _stdout_to_stderr():
            dist.init_process_group(backend="gloo" if backend == "rccl" else backend, rank=rank, world_size=world,
                                    timeout=to)
            dist.barrier()                      # (connections are made here at the latest)
    elif force and backend != "rccl" and not dist.is_initialized():
        import datetime
        with _stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=HOST_TIMEOUT_S))
    if backend == "rccl" and (world > 1 or force) and COMM is None:
        from . import rccl
        set_default_comm(rccl.init_comm(rank, world, torch.device("cuda", local)))
    return rank, local, world


def shutdown() -> None:
    """destroy the default communicator and the torch.distributed group (if any)"""
    global COMM
    if COMM is not None:
        COMM.destroy()
        COMM = None
    if dist.is_initialized():
        dist.destroy_process_group()
