"""Training-mode BatchNorm WITHOUT finalize launches (round 4; include/nsid.h "lazy" forms, csrc/nsid_common.h): the producing kernels
add their column sums into 64-bit fixed-point accumulators, the first consumer evaluates the layer in its prologue.

What must hold (reference semantics: nn.BatchNorm2d in training mode, torch_vertex.py:154,161, torch_nn.py:32, graph_encoder.py:45,75,77):
  * the evaluated scale / shift / mean / invstd / unbiased variance equal the finalize kernels' (fp64 combine of float partials) to
    ~1e-6 relative (the only difference: 2^-28 quantisation of each row tile's sum against fp32 rounding of it);
  * every consumer form (GEMM operand load, kNN, bn_apply; backward apply) gives what its eager form gives from those vectors;
  * totals are order-independent: repeated launches give identical bits;
  * a whole two-stream training step with the lazy forms equals the step with finalize launches (neighbour ids forced: kNN is
    discontinuous) within bf16 noise of the differing statistics, and really runs without finalize launches (launch counters)."""
import pytest
import torch

from synth import GRAFP_CFG, synth_randn, synth_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture()
def ops():
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops as o
    o.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    o.LAZY_BN = 3
    yield o
    o.STAT_ARENA.end()
    o.LAZY_BN = 0
    o.reset_tuning()
    o.set_gemm_precision("fp32")
    F_.set_activation_dtype("fp32")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def producer(ops, M, K, Nout, groups, tag):
    x = synth_randn(f"lz_x{tag}{M}{K}", M, groups * K).to(DEV).to(BF)
    w = (synth_randn(f"lz_w{tag}{Nout}{K}", groups * Nout, K) * K ** -0.5).to(DEV)
    bias = synth_randn(f"lz_b{tag}{Nout}", groups * Nout).to(DEV)
    gamma = (1 + 0.1 * synth_randn(f"lz_g{tag}", groups * Nout)).to(DEV)
    beta = (0.1 * synth_randn(f"lz_be{tag}", groups * Nout)).to(DEV)
    return x, w, bias, gamma, beta


@pytest.mark.parametrize("M,K,Nout,groups", [(65536, 64, 64, 1), (32768, 64, 64, 4), (16384, 256, 1024, 1), (8192, 2048, 512, 1),
                                             (640, 64, 128, 1)])
def test_fixed_point_statistics_equal_the_finalize_kernels(ops, M, K, Nout, groups):
    x, w, bias, gamma, beta = producer(ops, M, K, Nout, groups, "a")
    C = groups * Nout
    r0, stat = ops.linear_fwd(x, w, bias, M, Nout, K, groups, want_stat=True)
    eager, uvar0 = ops.bn_finalize_deferred(stat, M, gamma, beta)
    R = ops.stat_replicas(M)
    outs = []
    for rep in range(3):
        acc = torch.zeros(R * 2 * C, device=DEV, dtype=torch.int64)
        r1, _ = ops.linear_fwd(x, w, bias, M, Nout, K, groups, stat_acc=(acc, R))
        assert torch.equal(r1, r0)
        lazy, uvar1 = ops.lazy_affine(acc, R, M, gamma, beta)
        assert lazy.lazy.pending
        lazy.materialize()
        assert not lazy.lazy.pending
        outs.append((acc.clone(), lazy.lazy.out.clone()))
    for a, o in outs[1:]:                       # integer totals: identical whatever order the workgroups arrived in
        assert torch.equal(a, outs[0][0]) and torch.equal(o, outs[0][1])
    for name, e, l in (("scale", eager.scale, lazy.scale), ("shift", eager.shift, lazy.shift), ("mean", eager.mean, lazy.mean),
                       ("invstd", eager.invstd, lazy.invstd), ("uvar", uvar0, uvar1)):
        assert rel(l, e) < 2e-6, (name, rel(l, e))


def test_consumers_evaluate_the_layer_in_their_prologue(ops):
    """GEMM operand load (+ReLU), bn_apply (+residual), kNN: each with a pending layer, against the eager form on the evaluated vectors"""
    B, N, C = 32, 256, 64
    M = B * N
    x, w, bias, gamma, beta = producer(ops, M, C, C, 1, "c")
    R = ops.stat_replicas(M)

    def fresh():
        acc = torch.zeros(R * 2 * C, device=DEV, dtype=torch.int64)
        r, _ = ops.linear_fwd(x, w, bias, M, C, C, 1, stat_acc=(acc, R))
        return r, ops.lazy_affine(acc, R, M, gamma, beta)[0]

    r, ref = fresh()
    ref.materialize()
    before = ops.launch_counters(reset=True)
    w2 = (synth_randn("lz_w2", 2 * C, C) * C ** -0.5).to(DEV)
    # (a) bn_apply with a residual
    r, aff = fresh()
    out = ops.bn_apply(r, aff, ops.ACT_LEAKY, residual=x)
    assert not aff.lazy.pending and torch.equal(aff.lazy.out, ref.lazy.out)
    assert torch.equal(out, ops.bn_apply(r, ref, ops.ACT_LEAKY, residual=x))
    # (b) the next GEMM's operand load (BatchNorm + ReLU), its own statistics added in fixed point as well
    r, aff = fresh()
    acc2 = torch.zeros(R * 2 * 2 * C, device=DEV, dtype=torch.int64)
    y, _ = ops.linear_fwd(r, w2, None, M, 2 * C, C, 1, act_in=ops.ACT_RELU, in_aff=aff, stat_acc=(acc2, R))
    assert not aff.lazy.pending and torch.equal(aff.lazy.out, ref.lazy.out)
    y_ref, _ = ops.linear_fwd(r, w2, None, M, 2 * C, C, 1, ref.scale, ref.shift, ops.ACT_RELU)
    assert torch.equal(y, y_ref)
    # (c) the kNN graph
    r, aff = fresh()
    idx = ops.knn_graph(r, B, N, C, 3, 1, aff)
    assert not aff.lazy.pending and torch.equal(aff.lazy.out, ref.lazy.out)
    assert torch.equal(idx, ops.knn_graph(r, B, N, C, 3, 1, ref))
    cnt = ops.launch_counters()
    assert cnt["bn_lazy_finalize"] == 3 and cnt["bn_materialize"] == 0 and cnt["bn_stat_acc"] >= 4, cnt


@pytest.mark.parametrize("M,C,act,from_gemm", [(65536, 64, 0, False), (16384, 256, 1, False), (8192, 2048, 1, True), (640, 128, 2, False)])
def test_backward_sums_in_fixed_point(ops, M, C, act, from_gemm):
    """BatchNorm backward: reduce (stand-alone or in the backward-data GEMM's epilogue) into accumulators + apply with the prologue,
    against reduce + finalize + apply"""
    r = synth_randn(f"lzb_r{M}{C}", M, C).to(DEV).to(BF)
    gamma = (1 + 0.1 * synth_randn("lzb_g", C)).to(DEV)
    beta = (0.1 * synth_randn("lzb_b", C)).to(DEV)
    mean = r.float().mean(0)
    invstd = (r.float().var(0, unbiased=False) + 1e-5).rsqrt()
    sc = gamma * invstd
    buf = torch.stack([sc, beta - mean * sc, mean, invstd, torch.ones_like(mean)]).contiguous()
    eager = ops.BNAffine(buf[0], buf[1], buf[2], buf[3])
    lazy = ops.BNAffine(buf[0], buf[1], buf[2], buf[3], lazy=ops.LazyStats(None, 1, C, M, gamma, beta, 1e-5, buf))
    lazy.lazy.pending = False
    if from_gemm:
        Nout = 512
        dy = (synth_randn(f"lzb_dy{M}", M, Nout) * 1e-3).to(DEV).to(BF)
        w = (synth_randn("lzb_w", Nout, C) * C ** -0.5).to(DEV)
        dout0, part0 = ops.linear_bwd_data(dy, w, M, Nout, C, bn=(r, eager, act))
        dout1, part1 = ops.linear_bwd_data(dy, w, M, Nout, C, bn=(r, lazy, act))
        assert torch.equal(dout0, dout1) and isinstance(part1, ops.BwdSums) and torch.is_tensor(part0)
    else:
        dout0 = dout1 = (synth_randn(f"lzb_d{M}{C}", M, C) * 1e-3).to(DEV).to(BF)
        part0 = part1 = None
    dg0, db0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dr0 = ops.bn_backward(dout0, r, eager, act, dg0, db0, partial=part0)
    dr1 = ops.bn_backward(dout1, r, lazy, act, dg1, db1, partial=part1)
    assert rel(dg1, dg0) < 1e-5 and rel(db1, db0) < 1e-5, (rel(dg1, dg0), rel(db1, db0))
    # dr is bf16: equal up to one rounding where the coefficients differ in their last bits
    assert float((dr1.float() - dr0.float()).abs().max()) <= 2 ** -8 * float(dr0.float().abs().max())
    assert float((dr1 != dr0).float().mean()) < 1e-3


def build(k=3, overlap=True):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=8, k=k, size="t"), overlap_views=overlap)


def test_two_stream_step_without_finalize_launches(ops):
    """one contrastive step (train.py:53-75) on two streams, bf16 storage: lazy forms against finalize launches, neighbour ids of the
    first run forced in the second; the running statistics after the deferred update included"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    B = 16
    gi, gj = torch.Generator().manual_seed(0), torch.Generator().manual_seed(1)
    x_i = torch.randn(B, 64, 128, generator=gi) * 20 - 40
    x_j = (x_i + 3 * torch.randn(B, 64, 128, generator=gj)).to(DEV)
    x_i = x_i.to(DEV)
    res = {}
    tape = None
    ops.set_tuning("ws_gemm", 0)        # both paths on the tile GEMMs: the fixed-point statistics forms exist there only
    for mode in (True, False):
        ops.LAZY_BN = 3 if mode else 0
        torch.manual_seed(42)
        model = build()
        model.load_state_dict(synth_state(model.state_dict()))
        model.to(DEV).train()
        opt = FusedClipAdam(model.parameters(), lr=8e-5, max_norm=1.0)
        opt.zero_grad()
        ops.launch_counters(reset=True)
        try:
            F_.TAPE = F_.KnnTape(replay=tape)
            _, _, z_i, z_j = model(x_i, x_j)
            if tape is None:
                tape = [t.clone() for t in F_.TAPE.recorded]
        finally:
            F_.TAPE = None
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        F_.join_side_streams()
        torch.cuda.synchronize()
        cnt = ops.launch_counters()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        stats = {n: b.detach().clone() for n, b in model.named_buffers() if b.dtype.is_floating_point and "relative_pos" not in n}
        res[mode] = (float(loss.detach()), grads, stats, cnt)
    (l1, g1, s1, c1), (l0, g0, s0, c0) = res[True], res[False]
    assert c1["bn_materialize"] == 0 and c1["bn_lazy_finalize"] >= 2 * 2 * 64 and c1["bn_stat_acc"] >= 2 * 2 * 64, c1
    assert c0["bn_lazy_finalize"] == 0 and c0["bn_stat_acc"] == 0, c0
    print("lazy vs finalize: dloss", abs(l1 - l0))
    assert abs(l1 - l0) < 2e-3 * max(1.0, abs(l0))
    for n in s0:
        assert rel(s1[n], s0[n]) < 1e-3, (n, rel(s1[n], s0[n]))
    num = sum(float(((g1[n] - g0[n]).double() ** 2).sum()) for n in g0)
    den = sum(float((g0[n].double() ** 2).sum()) for n in g0)
    print("lazy vs finalize: gradient rel L2", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 5e-2
