"""CPU, world_size = 2 over gloo: the data-parallel host path (neuralsampleid_amd/parallel.py) reproduces the
single-process global-batch step — loss, embedding gradients and parameter gradients — when each rank owns half of
the clips.  The collective plumbing under test is the product's; the per-rank loss rows come from the oracle here
(the HIP kernel is injected on the GPU, tests/test_ops_gpu.py::test_ntxent_sharded checks the same split there)."""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:           # spawned ranks re-import this module without conftest.py
        sys.path.insert(0, _p)

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_torch as R
from synth import GRAFP_CFG, synth_randn

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def oracle_rows(zi_all, zj_all, tau, p0, n):
    """same contract as ops.ntxent_fwd_bwd: (sum of owned rows / M, dz_i, dz_j for the owned pairs)"""
    with torch.enable_grad():          # called from inside autograd.Function.forward, where grad mode is off
        zi = zi_all.detach().clone().requires_grad_(True)
        zj = zj_all.detach().clone().requires_grad_(True)
        M = 2 * zi.shape[0]
        z = torch.stack((zi, zj), dim=1).reshape(M, -1)
        # d(global mean loss)/dz for the owned pairs needs the FULL loss (other ranks' rows see these columns too)
        full = R.ntxent(zi, zj, tau)
        full.backward()
    part = R.ntxent_rows(z.detach(), 2 * p0, 2 * n, tau) / M
    return part.reshape(1), zi.grad[p0:p0 + n].clone(), zj.grad[p0:p0 + n].clone()


def tiny_params():
    g = torch.Generator().manual_seed(5)
    return {"w1": torch.randn(24, 16, generator=g) * 0.3, "w2": torch.randn(8, 24, generator=g) * 0.3}


def embed(x, P):
    z = torch.tanh(x @ P["w1"].t()) @ P["w2"].t()
    return z / z.norm(dim=1, keepdim=True).clamp_min(1e-10)


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from neuralsampleid_amd import parallel
    r, local, world = parallel.init_from_env("gloo")
    assert (r, world) == (rank, WORLD)
    B = 6
    p0, n = parallel.shard_range(B, rank, world)
    x_i, x_j = synth_randn("par_xi", B, 16), synth_randn("par_xj", B, 16)
    P = {k: v.clone().requires_grad_(True) for k, v in tiny_params().items()}
    z_i, z_j = embed(x_i[p0:p0 + n], P), embed(x_j[p0:p0 + n], P)
    loss = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG, rows_fn=oracle_rows)
    loss.backward()
    flat = torch.cat([P[k].grad.reshape(-1) for k in sorted(P)])
    bucketed = flat.clone()
    parallel.allreduce_gradients(flat)                                  # one collective
    works = parallel.allreduce_gradients(bucketed, bucket_bytes=256, async_op=True)   # bucketed + async
    for w in works:
        w.wait()
    gz = parallel.gather_embeddings(z_i.detach())
    if rank == 0:
        out.put((float(loss.detach()), flat.tolist(), bucketed.tolist(), gz.tolist()))   # plain data: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    deadline = time.time() + 120
    while q.empty():
        assert time.time() < deadline and all(p.exitcode in (None, 0) for p in procs), \
            f"ranks died or hung: {[p.exitcode for p in procs]}"
        time.sleep(0.2)
    loss2, flat2, bucketed2, gz = q.get()
    flat2, bucketed2, gz = torch.tensor(flat2), torch.tensor(bucketed2), torch.tensor(gz)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    B = 6
    x_i, x_j = synth_randn("par_xi", B, 16), synth_randn("par_xj", B, 16)
    P = {k: v.clone().requires_grad_(True) for k, v in tiny_params().items()}
    z_i, z_j = embed(x_i, P), embed(x_j, P)
    loss1 = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
    loss1.backward()
    flat1 = torch.cat([P[k].grad.reshape(-1) for k in sorted(P)])
    assert abs(loss2 - float(loss1.detach())) < 1e-5
    assert torch.allclose(flat2, flat1, atol=1e-5, rtol=1e-4)          # SUM of per-rank grads == global-batch grad
    assert torch.allclose(bucketed2, flat1, atol=1e-5, rtol=1e-4)
    assert torch.allclose(gz, z_i.detach(), atol=1e-6)                  # rank-major gather == global pair order


def _model_state():
    """state of SimCLR(GraphEncoder 't', k=3) from the reference's key/shape list + the synthetic per-key rule"""
    import json
    from synth import synth_tensor
    with open(os.path.join(ROOT, "tests", "golden", "state_shapes.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    return {k: synth_tensor(k, torch.empty(s)) for k, s in shapes.items() if "relative_pos" not in k}


def _replica_forward(P, x_i, x_j):
    """one replica's forward on its shard: per-replica BatchNorm statistics (nn.DataParallel, train.py:117-120; no SyncBN)"""
    plan = R.encoder_plan("t", 3)
    _, _, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, True, R.BNState())
    return z_i, z_j


def _model_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuralsampleid_amd import parallel
    from synth import synth_clips
    parallel.init_from_env("gloo")
    Bg = 4
    p0, n = parallel.shard_range(Bg, rank, WORLD)
    x_i, x_j = synth_clips(Bg)
    P = _model_state()
    keys = R.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    z_i, z_j = _replica_forward(P, x_i[p0:p0 + n], x_j[p0:p0 + n])
    loss = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG, rows_fn=oracle_rows)      # z all-gather, global-batch negatives
    loss.backward()
    flat = torch.cat([P[k].grad.reshape(-1) for k in keys if P[k].grad is not None])
    parallel.allreduce_gradients(flat, bucket_bytes=16 << 20)                        # SUM over ranks, 16 MB buckets
    if rank == 0:
        chk = [float(flat.double().norm()), float(flat.double().sum()), float(flat[::100003].double().abs().sum())]
        out.put((float(loss.detach()), chk, flat[::4099].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_on_the_real_model_graph():
    """BASELINE config 3 on CPU: the GraFP model itself (the oracle's graph: 12 Grapher + FFN blocks, train-mode BatchNorm
    per replica) stepped by two gloo ranks through neuralsampleid_amd/parallel.py — embedding all-gather, rank-local NT-Xent
    rows over the global batch, bucketed SUM all-reduce of all 18.4 M gradients — equals the single-process evaluation
    of what nn.DataParallel computes (train.py:117-120: every replica normalises with its own batch statistics, the loss
    sees the gathered global batch, gradients are summed)."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    # single-process reference while the ranks run
    torch.set_num_threads(4)
    from synth import synth_clips
    Bg = 4
    x_i, x_j = synth_clips(Bg)
    P = _model_state()
    keys = R.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    zs = [_replica_forward(P, x_i[r * 2:(r + 1) * 2], x_j[r * 2:(r + 1) * 2]) for r in range(WORLD)]
    loss1 = R.ntxent(torch.cat([z[0] for z in zs]), torch.cat([z[1] for z in zs]), GRAFP_CFG["tau"])
    loss1.backward()
    flat1 = torch.cat([P[k].grad.reshape(-1) for k in keys if P[k].grad is not None])
    deadline = time.time() + 300
    while q.empty():
        assert time.time() < deadline and all(p.exitcode in (None, 0) for p in procs), \
            f"ranks died or hung: {[p.exitcode for p in procs]}"
        time.sleep(0.2)
    loss2, chk, sample = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert flat1.numel() == 18366856
    assert abs(loss2 - float(loss1.detach())) < 2e-5
    n1 = float(flat1.double().norm())
    assert abs(chk[0] - n1) / n1 < 2e-3, (chk[0], n1)                   # fp32 summation order only (two threads vs four)
    s1 = flat1[::4099]
    assert float((torch.tensor(sample) - s1).norm() / s1.norm()) < 2e-2      # early-layer gradients: the fp32 floor of DESIGN.md section 4


def _reducer_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank))
    from neuralsampleid_amd import parallel
    parallel.init_from_env("gloo")
    sizes = [40, 8, 300, 12, 64, 700]            # "parameters" in forward order
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) // 4 * 4
    flat = torch.zeros(tot)
    params = [flat[o:o + n] for o, n in zip(offs, sizes)]
    red = parallel.GradReducer(params, flat, offs, bucket_bytes=1024, uses_per_step=2)
    red.start_step()
    order = []
    for view in range(2):                        # two views; backward visits parameters last-to-first
        for i in reversed(range(len(params))):
            params[i] += (rank + 1) * (i + 1)    # this rank's gradient contribution
            before = len(red.fired)
            red.block_done([params[i]])
            if len(red.fired) > before:
                order.append((view, i, list(red.fired)))
    red.finish()
    if rank == 0:
        out.put((flat.tolist(), red.bounds, order, offs, sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_buckets_fire_early_and_sum():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    deadline = time.time() + 120
    while q.empty():
        assert time.time() < deadline and all(p.exitcode in (None, 0) for p in procs)
        time.sleep(0.2)
    flat, bounds, order, offs, sizes = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    flat = torch.tensor(flat)
    for i, (o, n) in enumerate(zip(offs, sizes)):            # sum over ranks (1+2) x two views
        assert torch.equal(flat[o:o + n], torch.full((n,), 2.0 * 3 * (i + 1)))
    assert bounds[0][1] == flat.numel() and bounds[-1][0] == 0 and len(bounds) >= 3
    assert all(view == 1 for view, _, _ in order)             # nothing fires before the second view contributed
    assert order[0][2] == [0] and order[0][1] > 0             # the LAST parameters' bucket fires first, mid-backward


def test_shard_range():
    from neuralsampleid_amd import parallel
    assert [parallel.shard_range(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 256)]
    with pytest.raises(ValueError):
        parallel.shard_range(10, 0, 4)


def test_single_process_paths_need_no_process_group():
    from neuralsampleid_amd import parallel
    g = torch.zeros(10)
    assert parallel.allreduce_gradients(g) == []
    assert parallel.shard_range(256, 0, 1) == (0, 256)


def _uid_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank))
    from neuralsampleid_amd import parallel, rccl
    parallel.init_from_env("gloo")
    seen = {}

    class FakeComm:                                  # stands in for ncclCommInitRank: no GPU in this container
        def __init__(self, rank, world, uid, device=None):
            seen.update(rank=rank, world=world, uid=uid)
    real, rccl.RcclComm = rccl.RcclComm, FakeComm
    try:
        rccl.init_comm(rank, WORLD)
    finally:
        rccl.RcclComm = real
    gathered = [None] * WORLD
    dist.all_gather_object(gathered, (seen["rank"], seen["world"], seen["uid"]))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_bootstrap_broadcasts_rank0_unique_id():
    """the direct-RCCL communicator's bootstrap: rank 0's 128-byte ncclUniqueId reaches every rank over gloo"""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    deadline = time.time() + 120
    while q.empty():
        assert time.time() < deadline and all(p.exitcode in (None, 0) for p in procs)
        time.sleep(0.2)
    got = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1] and all(g[1] == WORLD for g in got)
    assert len(got[0][2]) == 128 and got[0][2] == got[1][2]


def test_rccl_library_exports_the_bound_entry_points():
    from neuralsampleid_amd import rccl
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclAllGather", "ncclCommDestroy",
                 "ncclGetErrorString", "ncclCommGetAsyncError"):
        assert hasattr(L, name)
    assert rccl.version() >= 21800
    assert len(rccl.new_unique_id()) == rccl.NCCL_UNIQUE_ID_BYTES


def test_stacked_view_gather_keeps_rank_major_pair_order():
    """the single all-gather of both views (RCCL path) must hand NT-Xent the same row order as two separate gathers"""
    from neuralsampleid_amd import parallel
    world, B, d = 3, 4, 5
    zi = [torch.randn(B, d) for _ in range(world)]
    zj = [torch.randn(B, d) for _ in range(world)]
    gathered = torch.cat([torch.stack((a, b)) for a, b in zip(zi, zj)])            # what ncclAllGather returns
    gi, gj = parallel._split_views(gathered, world)
    assert torch.equal(gi, torch.cat(zi)) and torch.equal(gj, torch.cat(zj))


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks through torch.distributed.run (fresh
    child processes, before this process makes any GPU call). In this container there is no GPU, so the ranks die: the
    spawner must come back promptly with a non-zero exit code and no JSON line — never hang, never print a fake result."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NSID_BENCH_TIMEOUT_S"] = "240"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=280)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1
        return
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "spawning 2 ranks" in r.stderr and time.time() - t0 < 240
