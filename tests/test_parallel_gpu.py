"""GPU: the data-parallel code path (RCCL all-gather of z, sharded NT-Xent, bucketed all-reduce overlapped with backward)
on a 1-rank group must reproduce the plain single-process step — same loss, same gradients — i.e. no contribution
is lost to a bucket that fires early, and the collectives compose with the HIP kernels' atomically accumulated grads.
Both transports are covered: the direct RCCL communicator (neuralsampleid_amd/rccl.py, the product path — eager, with
two-stream views, and captured in a hipGraph) and torch.distributed's "nccl" group."""
import os

import pytest
import torch
import torch.distributed as dist

from synth import GRAFP_CFG, synth_clips

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def one_rank_group():
    old = {k: os.environ.get(k) for k in ("NSID_FORCE_COLLECTIVES", "MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(NSID_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()
    from neuralsampleid_amd import functional
    functional.GRAD_READY_HOOK = None
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.fixture()
def rccl_comm():
    from neuralsampleid_amd import functional, parallel, rccl
    old = os.environ.get("NSID_FORCE_COLLECTIVES")
    os.environ["NSID_FORCE_COLLECTIVES"] = "1"
    comm = rccl.init_comm(0, 1)
    parallel.set_default_comm(comm)
    yield comm
    torch.cuda.synchronize()
    parallel.set_default_comm(None)
    comm.destroy()
    functional.GRAD_READY_HOOK = None
    if old is None:
        os.environ.pop("NSID_FORCE_COLLECTIVES", None)
    else:
        os.environ["NSID_FORCE_COLLECTIVES"] = old


def build():
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    torch.manual_seed(11)
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=8, k=3, size="t")).to(DEV).train()


def test_reducer_step_equals_plain_step(one_rank_group):
    from neuralsampleid_amd import parallel
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    x_i, x_j = (t.to(DEV) for t in synth_clips(32))
    # plain
    m0 = build()
    o0 = FusedClipAdam(m0.parameters(), lr=8e-5)
    o0.zero_grad()
    _, _, z_i, z_j = m0(x_i, x_j)
    l0 = ntxent_loss(z_i, z_j, GRAFP_CFG)
    l0.backward()
    g0 = o0.flat_g.clone()
    # data-parallel path, small buckets so that many fire in the middle of backward
    m1 = build()
    o1 = FusedClipAdam(m1.parameters(), lr=8e-5)
    red = parallel.GradReducer(o1.params, o1.flat_g, o1.offsets, bucket_bytes=1 << 20).install()
    assert len(red.bounds) > 10
    o1.zero_grad()
    red.start_step()
    _, _, z_i, z_j = m1(x_i, x_j)
    l1 = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG)
    l1.backward()
    fired_during_backward = len(red.fired)
    red.finish()
    torch.cuda.synchronize()
    assert fired_during_backward == len(red.bounds)                   # every bucket fired from the hooks
    assert abs(float(l0) - float(l1)) < 1e-5
    rel = float((o1.flat_g - g0).norm() / g0.norm())
    assert rel < 1e-3, rel                                            # fp32 atomics order + kNN near-ties only


def test_rccl_binding_one_rank(rccl_comm):
    from neuralsampleid_amd import rccl
    assert rccl.version() >= 21800 and rccl_comm.world == 1 and rccl_comm.async_error() == 0
    x = torch.arange(1000, device=DEV, dtype=torch.float32)
    y = rccl_comm.all_reduce_(x.clone())
    g = rccl_comm.all_gather(x.reshape(10, 100))
    b = torch.ones(64, device=DEV, dtype=torch.bfloat16)
    rccl_comm.all_reduce_async_(b)
    rccl_comm.wait_async()
    torch.cuda.synchronize()
    assert torch.equal(y, x) and torch.equal(g, x.reshape(10, 100)) and torch.equal(b, torch.ones_like(b))
    with pytest.raises(ValueError):
        rccl_comm.all_reduce_(torch.zeros(4))                          # host tensor


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_reducer_step_equals_plain_step(rccl_comm, overlap):
    from neuralsampleid_amd import parallel
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    x_i, x_j = (t.to(DEV) for t in synth_clips(32))
    m0 = build()
    o0 = FusedClipAdam(m0.parameters(), lr=8e-5)
    o0.zero_grad()
    _, _, z_i, z_j = m0(x_i, x_j)
    l0 = ntxent_loss(z_i, z_j, GRAFP_CFG)
    l0.backward()
    torch.cuda.synchronize()
    g0 = o0.flat_g.clone()
    m1 = build()
    m1.overlap_views = overlap
    o1 = FusedClipAdam(m1.parameters(), lr=8e-5)
    red = parallel.GradReducer(o1.params, o1.flat_g, o1.offsets, bucket_bytes=1 << 20).install()
    calls0 = rccl_comm.calls
    o1.zero_grad()
    red.start_step()
    _, _, z_i, z_j = m1(x_i, x_j)
    if overlap:
        red.streams = [torch.cuda.current_stream(), m1._side_stream]
    l1 = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG)
    l1.backward()
    fired_during_backward = len(red.fired)
    red.finish()
    torch.cuda.synchronize()
    assert fired_during_backward == len(red.bounds)
    assert rccl_comm.calls - calls0 == len(red.bounds) + 2            # buckets + ONE all-gather of both views + the loss all-reduce
    assert abs(float(l0) - float(l1)) < 1e-5
    rel = float((o1.flat_g - g0).norm() / g0.norm())
    assert rel < 1e-3, rel


def test_rccl_step_captured_in_hipgraph(rccl_comm):
    """the whole step incl. the collectives on the communicator's stream is captured once and replayed: from the same
    state, one replay must produce the loss, the (all-reduced) gradients and the updated weights of one eager step"""
    from neuralsampleid_amd import parallel
    from neuralsampleid_amd.optim import FusedClipAdam
    x_i, x_j = (t.to(DEV) for t in synth_clips(16))
    m = build()
    m.overlap_views = True
    o = FusedClipAdam(m.parameters(), lr=8e-5)
    red = parallel.GradReducer(o.params, o.flat_g, o.offsets, bucket_bytes=4 << 20).install()
    loss_buf = torch.zeros((), device=DEV)

    def step():
        o.zero_grad()
        red.start_step()
        if m._side_stream is not None:
            red.streams = [torch.cuda.current_stream(), m._side_stream]
        _, _, z_i, z_j = m(x_i, x_j)
        loss = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        red.finish()
        o.step()
        loss_buf.copy_(loss.detach())

    def snapshot():
        return ({k: v.clone() for k, v in m.state_dict().items()}, o.exp_avg.clone(), o.exp_avg_sq.clone(),
                o.step_count.clone())

    def restore(snap):
        m.load_state_dict(snap[0])                                     # copies in place: captured pointers stay valid
        o.exp_avg.copy_(snap[1]); o.exp_avg_sq.copy_(snap[2]); o.step_count.copy_(snap[3])

    init = snapshot()
    step()                                                             # eager reference step (also the warm-up)
    torch.cuda.synchronize()
    ref = (float(loss_buf), o.flat_g.clone(), o.flat_p.clone(), snapshot()[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    calls = rccl_comm.calls
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    assert rccl_comm.calls - calls == len(red.bounds) + 2              # the collectives were enqueued under capture
    restore(init)
    loss_buf.zero_(); o.flat_g.fill_(7.0)
    graph.replay()
    torch.cuda.synchronize()
    assert int(o.step_count) == 1 and abs(float(loss_buf) - ref[0]) < 1e-5
    assert float((o.flat_g - ref[1]).norm() / ref[1].norm()) < 1e-3    # fp32 atomics order + kNN near-ties only
    assert float((o.flat_p - ref[2]).abs().max()) < 2.1 * 8e-5         # one Adam step: |dp| <= lr per weight
    assert float((o.flat_p - ref[2]).abs().mean()) < 0.05 * 8e-5
    for k, v in ref[3].items():
        if k.endswith(("running_mean", "running_var")):
            assert float((m.state_dict()[k] - v).abs().max()) < 1e-4, k


def _run_bench_rehearsal(n, extra):
    """the whole N-rank orchestration of bench.py on this one-GPU box: torch.distributed.run starts N fresh ranks that share the
    device, torch.distributed/gloo carries the data-path collectives (RCCL refuses two ranks on one GPU)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NSID_BENCH_REHEARSAL"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(n), "--no-cpu-baseline", "--no-roofline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, (len(lines), r.stdout[-500:])            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_orchestration_rehearsal_two_and_four_ranks():
    """`bench.py --gpus N` end to end before the driver's first multi-GPU run — sharded inputs, z all-gather +
    global NT-Xent, bucketed gradient all-reduce through the reducer hooks, barrier + max-over-ranks timing, ONE JSON line.
    (What only the 8-GPU node can show stays untested: ncclCommInitRank across processes, RCCL kernels in a captured graph.)"""
    for n in (2, 4):                                                   # at most 4 processes on the card (the box allows 6)
        d = _run_bench_rehearsal(n, ["--steps", "2", "--warmup", "1", "--batch", "16"])
        assert d["n_gpus"] == n and d["steps"] == 2 and d["scaling"] == "weak"
        assert d["config"]["global_batch"] == 16 * n and d["config"]["parallelism"] == f"dp{n}"
        assert "REHEARSAL" in d["config"]["collectives"] and d["config"]["hipgraph"] is False
        assert d["value"] > 0 and abs(d["value"] - 16 * n * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
        assert 0 < d["config"]["final_loss"] < 10
    d = _run_bench_rehearsal(2, ["--mode", "infer", "--clips", "4096", "--micro-batch", "1024", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "shard2" and d["value"] > 0


def _spawn_bench(env_extra, args, timeout=400):
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.spawn_ranks starts the ranks itself (fresh children; this parent never
    touches the GPU). Rehearsal transport (gloo, ranks sharing the card)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NSID_BENCH_REHEARSAL"] = "gloo"
    env.update(env_extra)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--no-roofline",
                        "--steps", "3", "--warmup", "1", "--batch", "16"] + args, capture_output=True, text=True, env=env, timeout=timeout)
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith("{")], time.time() - t0


def test_a_rank_killed_mid_step_fails_the_job_loudly():
    """VERDICT r3 task 6: rank 1 dies (SIGKILL) inside the timed loop. The surviving rank sits in a collective; the launcher must end
    it, and the spawning parent must come back NON-ZERO within its deadline and print NO JSON line (never a result from a broken job).
    NSID_DP_RETRY=0: no second attempt."""
    r, lines, dt = _spawn_bench({"NSID_TEST_KILL_RANK": "1:1", "NSID_DP_RETRY": "0", "NSID_BENCH_TIMEOUT_S": "240",
                                 "NSID_HOST_TIMEOUT_S": "60"}, [])
    assert r.returncode != 0, r.stderr[-1500:]
    assert not lines, lines
    assert dt < 240, dt


def test_the_spawner_retries_once_with_eager_collectives():
    """the first attempt ("graph") loses rank 1 mid-step; spawn_ranks then starts ONE fresh set of ranks with NSID_DP_GRAPH=0 and that
    attempt's line is the result, labelled as such"""
    import json
    r, lines, dt = _spawn_bench({"NSID_TEST_KILL_RANK": "1:1:graph", "NSID_BENCH_TIMEOUT_S": "500", "NSID_HOST_TIMEOUT_S": "60"}, [],
                                timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["dp_attempt"] == "eager-retry" and d["config"]["hipgraph"] is False and d["n_gpus"] == 2
    assert "retrying ONCE with eager collectives" in r.stderr
