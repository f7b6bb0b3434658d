"""The deferred weight-gradient phase (nsid_linear_bwd_weight_grouped, csrc/gemm.hip wgrad_grouped_kernel): many layers' weight
gradients in one launch per tile class, both views of a layer as two row segments of one problem.

Reference sites: the backward of every conv at encoder/gcn_lib/torch_vertex.py:152-162, encoder/graph_encoder.py:74-77,
encoder/gcn_lib/torch_nn.py:56 (autograd's dW = dY^T X); train.py:70-75 (nothing reads a gradient before clip + step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def bf16_mode():
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    yield
    ops.reset_tuning()
    ops.set_gemm_precision("fp32")
    F_.set_activation_dtype("fp32")


# (M per view, Nout, K, groups, affine + ReLU on x): every conv layer of one size-'t' encoder at a reduced batch, the stem, odd row
# counts (predicated class), a single-view problem
LAYERS = [(2048, 64, 64, 1, False), (2048, 32, 32, 4, False), (2048, 64, 128, 1, True), (2048, 256, 64, 1, False),
          (2048, 64, 256, 1, True), (1024, 128, 128, 1, False), (1024, 64, 64, 4, False), (1024, 128, 256, 1, True),
          (1024, 512, 128, 1, False), (1024, 128, 512, 1, True), (512, 256, 256, 1, False), (512, 128, 128, 4, False),
          (512, 256, 512, 1, True), (512, 1024, 256, 1, False), (512, 256, 1024, 1, True), (256, 512, 512, 1, False),
          (256, 256, 256, 4, False), (256, 512, 1024, 1, True), (256, 2048, 512, 1, False), (256, 512, 2048, 1, True),
          (2048, 64, 8, 1, False), (96, 64, 64, 1, True), (200, 128, 64, 1, False)]


def _make(M, N, K, G, aff, seed, views=2):
    g = torch.Generator().manual_seed(seed)
    segs = []
    for v in range(views):
        dout = (0.5 * torch.randn(M, G * N, generator=g)).to(torch.bfloat16)
        x = torch.randn(M, G * K, generator=g).to(torch.bfloat16)
        sc = (1 + 0.2 * torch.randn(G * K, generator=g)) if aff else None
        sh = (0.3 * torch.randn(G * K, generator=g)) if aff else None
        segs.append((dout, x, sc, sh))
    return segs


def _ref(segs, N, K, G, aff):
    """fp64 of what the kernel computes: x as the MFMA sees it (affine + ReLU in fp32, rounded to bf16), dout as stored"""
    dw = torch.zeros(G * N, K, dtype=torch.float64)
    for dout, x, sc, sh in segs:
        xf = x.float()
        if aff:
            # the kernel's operand: one fused multiply-add in fp32 (= the fp64 sum rounded once), ReLU, one rounding to bf16
            xf = torch.relu((x.double() * sc.double() + sh.double()).float()).to(torch.bfloat16).float()
        for gi in range(G):
            dw[gi * N:(gi + 1) * N] += dout[:, gi * N:(gi + 1) * N].double().t() @ xf[:, gi * K:(gi + 1) * K].double()
    return dw


def test_grouped_weight_gradients_match_fp64_and_the_per_layer_launches(bf16_mode):
    from neuralsampleid_amd import ops
    items, refs, outs, singles = [], [], [], []
    for li, (M, N, K, G, aff) in enumerate(LAYERS):
        views = 1 if li == 3 else 2
        segs = _make(M, N, K, G, aff, 100 + li, views)
        refs.append(_ref(segs, N, K, G, aff))
        dw = torch.zeros(G * N, K, device=DEV)
        dw1 = torch.zeros(G * N, K, device=DEV)
        outs.append(dw)
        singles.append(dw1)
        for dout, x, sc, sh in segs:
            d, xx = dout.to(DEV), x.to(DEV)
            scd, shd = (sc.to(DEV), sh.to(DEV)) if aff else (None, None)
            act = ops.ACT_RELU if aff else ops.ACT_NONE
            items.append((d, xx, dw, M, N, K, G, scd, shd, act))
            ops.linear_bwd_weight(d, xx, dw1, M, N, K, G, scd, shd, act)
    # the two views of a layer arrive interleaved with other layers, as in backward
    order = list(range(0, len(items), 2)) + list(range(1, len(items), 2))
    ops.launch_counters(reset=True)
    ops.linear_bwd_weight_batch([items[i] for i in order])
    torch.cuda.synchronize()
    cnt = ops.launch_counters()
    assert 1 <= cnt["wgrad_grouped"] <= 6 and cnt["wgrad_grouped_w3"] >= 1 and cnt["gemm_bwd_weight"] == 0, cnt
    # a capped launch (max_workgroups: every workgroup walks several items with a static stride) adds the same sums once more
    first = [dw.clone() for dw in outs]
    ops.linear_bwd_weight_batch([items[i] for i in order], max_workgroups=64)
    torch.cuda.synchronize()
    for dw, f0 in zip(outs, first):
        assert float((dw - 2 * f0).abs().max()) <= 2e-5 * float(f0.abs().max()) + 1e-6
        dw.copy_(f0)
    for (M, N, K, G, aff), dw, dw1, ref in zip(LAYERS, outs, singles, refs):
        scale = float(ref.abs().max())
        e = float((dw.double().cpu() - ref).abs().max()) / scale
        e1 = float((dw1.double().cpu() - ref).abs().max()) / scale
        # fp32 accumulation of bf16 products over <= 4096 rows: 1e-5 relative to the largest entry (with the affine on the operand the
        # fp64 restatement rounds an operand to the other bf16 neighbour now and then: measured 5.4e-5 for BOTH forms); the grouped
        # launch and the per-layer launch differ by the order of their fp32 atomics only
        d01 = float((dw - dw1).abs().max()) / scale
        assert e < (2e-4 if aff else 2e-5) and e < 1.5 * e1 + 2e-6 and d01 < 1e-5, ((M, N, K, G, aff), e, e1, d01)


def test_grouped_weight_gradients_at_the_timed_batch(bf16_mode):
    """the problems of the C = 256 / 512 stages at B = 256 (M = 16 384 / 8 192 rows per view): splits per view = M / wgg_rows, the
    second view's rows in the second half of the splits; two settings of the split target give the same sums up to fp32 atomics order"""
    from neuralsampleid_amd import ops
    res = {}
    for rows in (8192, 2048):
        ops.set_tuning("wgg_rows", rows)
        items, outs = [], []
        for li, (M, N, K, G, aff) in enumerate([(16384, 1024, 256, 1, False), (16384, 256, 1024, 1, True), (8192, 256, 256, 4, False),
                                                 (8192, 2048, 512, 1, False)]):
            segs = _make(M, N, K, G, aff, 300 + li)
            dw = torch.zeros(G * N, K, device=DEV)
            outs.append(dw)
            for dout, x, sc, sh in segs:
                items.append((dout.to(DEV), x.to(DEV), dw, M, N, K, G, sc.to(DEV) if aff else None, sh.to(DEV) if aff else None,
                              ops.ACT_RELU if aff else ops.ACT_NONE))
            if rows == 8192:
                res[li] = _ref(segs, N, K, G, aff)
        ops.linear_bwd_weight_batch(items)
        torch.cuda.synchronize()
        for li, dw in enumerate(outs):
            e = float((dw.double().cpu() - res[li]).abs().max()) / float(res[li].abs().max())
            assert e < (2e-4 if li == 1 else 3e-5), (rows, li, e)


def test_grouped_downsample_and_fp32_head_problems(bf16_mode):
    """the two other members of the deferred phase: a Downsample's packed weight gradient (x read as the zero-padded 3-tap view,
    encoder/graph_encoder.py:44-50) and the projector head's fp32 tensors (simclr/simclr.py:23-28), each with both views as segments,
    against the per-layer launches"""
    from neuralsampleid_amd import ops
    g = torch.Generator().manual_seed(5)
    items, refs = [], []
    for (B, N, C, Co) in ((8, 256, 64, 128), (8, 64, 256, 512)):
        Mo = B * N // 2
        dwp, ref = torch.zeros(Co, 3 * C, device=DEV), torch.zeros(Co, 3 * C, device=DEV)
        grad, grad_ref = torch.zeros(Co, C, 3, 3, device=DEV), torch.zeros(Co, C, 3, 3, device=DEV)
        for v in range(2):
            dr = (0.5 * torch.randn(Mo, Co, generator=g)).to(torch.bfloat16).to(DEV)
            x = torch.randn(B * N, C, generator=g).to(torch.bfloat16).to(DEV)
            items.append((dr, x, dwp, Mo, Co, 3 * C, 1, None, None, ops.ACT_NONE, ("ds", B, N, C, grad)))
            ops.downsample3_bwd_weight(dr, x, ref, B, N, C, Co)
        ops.unpack_ds_wgrad(ref, grad_ref)
        refs.append((grad, grad_ref))
    for (M, N, K) in ((256, 1024, 512), (256, 128, 4096)):
        dw, ref = torch.zeros(N, K, device=DEV), torch.zeros(N, K, device=DEV)
        for v in range(2):
            d = (0.5 * torch.randn(M, N, generator=g)).to(DEV)
            x = torch.randn(M, K, generator=g).to(DEV)
            items.append((d, x, dw, M, N, K, 1, None, None, ops.ACT_NONE, None))
            ops.linear_bwd_weight(d, x, ref, M, N, K)
        refs.append((dw, ref))
    ops.launch_counters(reset=True)
    ops.linear_bwd_weight_batch(items)
    torch.cuda.synchronize()
    cnt = ops.launch_counters()
    assert cnt["wgrad_grouped"] == 2 and cnt["gemm_bwd_weight"] == 0, cnt
    for out, ref in refs:
        scale = float(ref.abs().max())
        assert scale > 0 and float((out - ref).abs().max()) / scale < 1e-5, (tuple(out.shape), float((out - ref).abs().max()), scale)


def test_grouped_entry_refuses_bad_arguments(bf16_mode):
    import ctypes
    from neuralsampleid_amd import ops
    from neuralsampleid_amd._lib import WgradProblem, lib
    d = torch.zeros(128, 64, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(64, 64, device=DEV)
    q = WgradProblem()
    q.dout[0], q.x[0], q.dw, q.ldd, q.ldx, q.M, q.Nout, q.K, q.groups = d.data_ptr(), d.data_ptr(), dw.data_ptr(), 64, 64, 128, 64, 64, 1
    arr = (WgradProblem * 1)(q)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.nsid_linear_bwd_weight_grouped(ctypes.addressof(arr), 1, 7, 0, s) == -1            # no such storage type
    assert lib.nsid_linear_bwd_weight_grouped(ctypes.addressof(arr), 0, ops.BF16, 0, s) == -1
    arr[0].x[0] = None
    assert lib.nsid_linear_bwd_weight_grouped(ctypes.addressof(arr), 1, ops.BF16, 0, s) == -1
    arr[0].x[0] = d.data_ptr()
    arr[0].dout[1] = d.data_ptr()                       # a second dout without a second x
    assert lib.nsid_linear_bwd_weight_grouped(ctypes.addressof(arr), 1, ops.BF16, 0, s) == -1
    arr[0].dout[1] = None
    assert lib.nsid_linear_bwd_weight_grouped(ctypes.addressof(arr), 1, ops.BF16, 0, s) == 0
    torch.cuda.synchronize()


def test_deferred_phase_gives_the_gradients_of_the_in_chain_launches(bf16_mode):
    """one B = 8 contrastive step with functional.DEFER_WGRAD on and off (neighbour ids forced to the first run's). (i) p.grad is
    complete when loss.backward() returns (the engine callback has flushed) and every deferred layer's gradient equals the per-layer
    launches evaluated at flush time from the recorded tensors (nothing the phase reads was overwritten meanwhile); (ii) against a
    step without deferral the conv weight gradients agree to the run-to-run noise of the step itself"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    from synth import GRAFP_CFG, synth_clips, synth_state
    x_i, x_j = synth_clips(8)
    grads, tape = {}, None
    for tag, defer, overlap, lanes in (("base", 0, False, 1), ("base2", 0, False, 1), ("defer", 1, False, 1), ("defer2s", 1, True, 1),
                                       ("defer2s_one_lane", 1, True, 0)):
        model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t"), overlap_views=overlap)
        model.load_state_dict(synth_state(model.state_dict()))
        model.to(DEV).train()
        opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
        defer_keep, F_.DEFER_WGRAD = F_.DEFER_WGRAD, defer
        lanes_keep, F_.DEFER_TWO_LANES = F_.DEFER_TWO_LANES, lanes
        F_.DEFERRED.verify = [] if defer else None
        F_.TAPE = F_.KnnTape(replay=tape)
        ops.launch_counters(reset=True)
        try:
            opt.zero_grad()
            _, _, z_i, z_j = model(x_i.to(DEV), x_j.to(DEV))
            ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
            if tape is None:
                tape = [t.clone() for t in F_.TAPE.recorded]
            checks = F_.DEFERRED.verify
        finally:
            F_.TAPE = None
            F_.DEFER_WGRAD = defer_keep
            F_.DEFER_TWO_LANES = lanes_keep
            F_.DEFERRED.verify = None
        cnt = ops.launch_counters()
        assert (cnt["wgrad_grouped"] > 0) == bool(defer), cnt
        assert not F_.DEFERRED.items and not F_.DEFERRED.armed
        torch.cuda.synchronize()
        if defer:
            assert len(checks) == 67, len(checks)           # 12 blocks x 5 conv layers + the stem + 3 Downsamples + proj + projector
            for dw, t in checks:
                scale = max(float(t.abs().max()), 1e-6)
                assert float((dw - t).abs().max()) / scale < 2e-5, (tuple(dw.shape), float((dw - t).abs().max()), scale)
        grads[tag] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    # across runs: the aggregation backward rounds dy in arrival order (LDS atomics), so two steps differ by bf16-level noise that
    # train-mode BatchNorm at batch 8 spreads over every earlier layer: measured 0 ... 0.021 relative L2 on the stem weight between two
    # UNDEFERRED runs (tools/deferred_debug2.py). The exact statement is the flush-time check above; here: no conv weight further than
    # 0.08 from the undeferred step, the global norm within 2 %
    for tag in ("defer", "defer2s", "defer2s_one_lane"):
        for n, gr in grads[tag].items():
            if not n.endswith("0.weight") or "projector" in n:
                continue
            ref = grads["base"][n]
            assert float((gr - ref).norm()) / max(float(ref.norm()), 1e-9) < 0.08, (tag, n)
        gn = lambda d: float(torch.sqrt(sum(v.double().pow(2).sum() for v in d.values())))
        assert abs(gn(grads[tag]) - gn(grads["base"])) / gn(grads["base"]) < 0.02, tag


def test_the_phase_in_pieces_reports_every_block_after_its_gradients(bf16_mode):
    """data parallelism (functional.DEFER_CHUNKS = 3, parallel.GradReducer.install): the deferred phase goes out in pieces, late layers
    first, and a block reports its parameters (GRAD_READY_HOOK: the reducer fires the bucket's all-reduce there) only when the piece that
    holds its last gradient has been issued. A fake hook snapshots the reported gradients AT the report (stream order = what an
    all-reduce enqueued there would read): they must equal the final ones, every trainable parameter reports once per view, and the
    first report comes from the late layers."""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    from synth import GRAFP_CFG, synth_clips, synth_state
    x_i, x_j = synth_clips(8)
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t"), overlap_views=True)
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).train()
    opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
    names = {id(p): n for n, p in model.named_parameters()}
    reports, snaps = [], []

    def hook(params):
        F_.join_side_streams()                  # what the reducer's comm stream does: wait for every producer stream
        for p in params:
            reports.append(names[id(p)])
            snaps.append((names[id(p)], p.grad.detach().clone()))
    keep = (F_.GRAD_READY_HOOK, F_.DEFER_CHUNKS)
    F_.GRAD_READY_HOOK, F_.DEFER_CHUNKS = hook, 3
    try:
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i.to(DEV), x_j.to(DEV))
        ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
    finally:
        F_.GRAD_READY_HOOK, F_.DEFER_CHUNKS = keep
    torch.cuda.synchronize()
    final = {n: p.grad.detach() for n, p in model.named_parameters() if p.requires_grad}
    count = {}
    for n in reports:
        count[n] = count.get(n, 0) + 1
    assert set(count) == set(final) and set(count.values()) == {2}, {n: c for n, c in count.items() if c != 2}
    assert reports[0].startswith(("projector", "encoder.proj", "encoder.backbone.14")), reports[0]
    last = {}
    for n, g in snaps:                            # the SECOND report of a parameter (both views in) carries its complete gradient
        last[n] = g
    for n, g in last.items():
        assert torch.equal(g, final[n]), n
