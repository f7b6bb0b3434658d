"""Parity of the arithmetic bench.py TIMES (BASELINE config 2): bf16 activation storage + bf16 MFMA operands + fp32
accumulation, in TRAINING mode, against the reference's fp32 goldens — and of the exact bench configuration as a
combination (bf16 storage, two-stream views, BwdChain cross-block links, deferred running statistics,
FusedClipAdam(direct_grads), whole-step hipGraph, B = 256).

kNN is discontinuous, so — as everywhere else in this suite — the neighbour indices are teacher-forced to the ones the
compared-with run produced (functional.KnnTape); step 0 of train.py:53-75 is then a deterministic function of the inputs.

Tolerances are <= 3x the spread measured on MI355X (recorded next to each assert; the run writes what it measured to
gpurun_out/timed_arithmetic_measured.json)."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, ROOT
from compare import maxerr, relerr
from synth import GRAFP_CFG, synth_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEASURED = {}


def note(key, value):
    MEASURED[key] = value
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "timed_arithmetic_measured.json"), "w") as f:
            json.dump(MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture()
def bf16_mode():
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    yield
    F_.TAPE = None
    ops.set_gemm_precision("fp32")
    F_.set_activation_dtype("fp32")


def tape_of(g, tag):
    n = len([k_ for k_ in g if k_.startswith(f"knn.{tag}.")])
    return [g.t(f"knn.{tag}.{c}") for c in range(n)]


def build(k, overlap=False):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=k, size="t"),
                  overlap_views=overlap)


def emulated_step0(g, k):
    """step 0 of train.py:53-75 by the ORACLE with STORAGE = "bf16": the reference's graph with a bf16 rounding at every
    point where the MI355X path stores an activation / stages a GEMM operand in bf16 (oracle/ref_torch.py)"""
    import torch
    from oracle import ref_torch as R
    torch.set_num_threads(8)
    with open(os.path.join(GOLDEN, "state_shapes.json")) as f:
        shapes = {k_: tuple(v) for k_, v in json.load(f).items()}
    from synth import synth_tensor
    P = {k_: synth_tensor(k_, torch.empty(s_)) for k_, s_ in shapes.items()}
    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    R.STORAGE = "bf16"
    R.TAPE = R.KnnTape(replay=tape_of(g, "s0"))
    try:
        st = R.BNState()
        h_i, h_j, z_i, z_j = R.simclr_forward(g.t("x_i"), g.t("x_j"), P, GRAFP_CFG, R.encoder_plan("t", k), True, st)
        loss = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
        loss.backward()
    finally:
        R.STORAGE = None
        R.TAPE = None
    grads = {k_: P[k_].grad for k_ in keys if P[k_].grad is not None}
    gn = float(torch.sqrt(sum(v.double().pow(2).sum() for v in grads.values())))
    return dict(h_i=h_i.detach(), h_j=h_j.detach(), z_i=z_i.detach(), z_j=z_j.detach(), loss=float(loss.detach()),
                grads=grads, gnorm=gn, bn=st.updates)


@pytest.mark.parametrize("k", [3, 5])
def test_bf16_train_step_vs_oracle_and_goldens(bf16_mode, golden, k):
    """Step 0 of train.py:53-75 in the TIMED arithmetic (bf16 storage, bf16 MFMA operands, fp32 accumulate, FusedClipAdam
    with direct gradient accumulation), reference kNN indices forced, against

      (a) the oracle evaluated with the same rounding points (oracle.ref_torch.STORAGE = "bf16") — the tight check: both
          sides round the same tensors, they differ by summation order and by the bf16 rounding of GRADIENT tensors, which
          autograd on the oracle does not emulate;
      (b) the reference's fp32 goldens — the statement of what bf16 costs. Train-mode BatchNorm at batch 8 amplifies a
          rounding of 2^-9 per tensor to ~17 % on h (0.8 % in eval mode, tests/test_bf16_storage_gpu.py); the oracle's
          emulation deviates from the goldens by the same amount, and the HIP path must not deviate more than it does."""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    g = golden(f"e2e_b8_k{k}")
    with open(os.path.join(GOLDEN, f"e2e_b8_k{k}_checksums.json")) as f:
        chk = json.load(f)
    em = emulated_step0(g, k)
    model = build(k)
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).train()
    opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)     # direct gradient accumulation, as timed
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    F_.TAPE = F_.KnnTape(replay=tape_of(g, "s0"))
    opt.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    F_.TAPE = None
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None and p.requires_grad}
    opt.step()
    torch.cuda.synchronize()
    cosmin = lambda a, b: float(torch.nn.functional.cosine_similarity(a.detach().cpu(), b, dim=1).min())
    # ---- (a) against the emulation
    a = {"rel_h": max(relerr(h_i, em["h_i"]), relerr(h_j, em["h_j"])),
         "max_z": max(maxerr(z_i, em["z_i"]), maxerr(z_j, em["z_j"])),
         "cos_z_min": min(cosmin(z_i, em["z_i"]), cosmin(z_j, em["z_j"])),
         "dloss": abs(float(loss.detach()) - em["loss"]),
         "gnorm_rel": abs(float(opt.grad_norm) - em["gnorm"]) / em["gnorm"]}
    rel_g = {n: relerr(grads[n], v) for n, v in em["grads"].items() if float(v.norm()) > 1e-5}
    a["grad_rel_worst"] = max(rel_g.values())
    a["grad_rel_worst_name"] = max(rel_g, key=rel_g.get)
    a["grad_rel_median"] = sorted(rel_g.values())[len(rel_g) // 2]
    a["grad_rel_late"] = max(v for n, v in rel_g.items() if n.startswith(("encoder.backbone.14", "encoder.proj", "projector")))
    sd = model.state_dict()
    a["bn_stat_worst"] = max(maxerr(sd[n].float(), v.float()) / max(1.0, float(v.abs().max()))
                             for n, v in em["bn"].items() if not n.endswith("num_batches_tracked"))
    # ---- (b) against the fp32 goldens, next to the emulation's own deviation from them
    def dev(hi, zi, ls, gr):
        full = {n[5:]: relerr(gr[n[5:]], g.t(n)) for n in g if n.startswith("grad.") and float(g.t(n).norm()) > 1e-5}
        return {"rel_h": relerr(hi, g.t("h_i_train")), "cos_z_min": cosmin(zi, g.t("z_i_train")),
                "dloss": abs(ls - float(g["losses"][0])), "full_grads": full}
    b_hip = dev(h_i, z_i, float(loss.detach()), grads)
    b_em = dev(em["h_i"], em["z_i"], em["loss"], em["grads"])
    note(f"bf16_train_k{k}", {"vs_emulation": a, "hip_vs_golden": b_hip, "emulation_vs_golden": b_em})
    print("measured", json.dumps({"vs_emulation": a, "hip_vs_golden": b_hip, "emulation_vs_golden": b_em}, indent=1))
    # (a) measured on MI355X (k = 3 / 5): see the numbers recorded below each bound
    assert a["rel_h"] < TOL["rel_h"] and a["max_z"] < TOL["max_z"] and a["cos_z_min"] > TOL["cos_z_min"]
    assert a["dloss"] < TOL["dloss"] and a["gnorm_rel"] < TOL["gnorm_rel"]
    assert a["bn_stat_worst"] < TOL["bn_stat_worst"]
    # per-parameter gradients at batch 8 are recorded, not bounded (bounds of 3x the measured 0.65 / 0.40 / 0.34 relative
    # L2 cannot fail): the per-parameter statement of the timed arithmetic is made where train-mode BatchNorm does not amplify a
    # rounding 20-40x — one block deep (test_block_train_bf16_vs_emulation, four shapes) and at the timed batch of 256
    # (tests/test_b256_gpu.py)
    # (b) no further from the reference than the emulated arithmetic is
    assert b_hip["rel_h"] < 1.3 * b_em["rel_h"] + 1e-3 and b_hip["dloss"] < 1.5 * b_em["dloss"] + 2e-2
    assert b_hip["cos_z_min"] > 1.0 - 1.5 * (1.0 - b_em["cos_z_min"]) - 1e-3
    for n, e in b_hip["full_grads"].items():
        assert e < 1.3 * b_em["full_grads"][n] + 0.05, (n, e, b_em["full_grads"][n])


# (a)-bounds: <= 3x the values measured on MI355X in round 2 (k = 3 / 5): rel_h 0.062 / 0.067, |dz| 0.017 / 0.017,
# min cos z 0.9977 / 0.9976, |dloss| 0.013 / 0.010, global gradient norm 0.3 % / 0.07 %, per-parameter gradient relative L2:
# late layers 0.34 / 0.32, median 0.39 / 0.40, worst 0.65 / 0.64, running statistics 0.8 % / 0.9 %.
# Why the gradients are loose even against the emulation: the max-relative aggregation routes each gradient to ONE of k
# near-identical neighbours and ReLU masks gate the rest; a one-ulp difference in a bf16-stored activation flips such a
# choice, and train-mode BatchNorm at batch 8 then spreads it over every clip. The global norm (0.3 %) and the loss are
# the well-conditioned quantities; test_block_train_bf16_vs_emulation states the per-block (un-amplified) agreement.
TOL = {"rel_h": 0.15, "max_z": 0.05, "cos_z_min": 0.993, "dloss": 0.04, "gnorm_rel": 0.12, "bn_stat_worst": 0.025}
# (gnorm_rel is the difference of two global norms, a signed noise variable: 0.0011 / 0.0026 with the tile GEMMs of rounds 1-4. Round 5
# measured what ANOTHER correct rounding realisation does to it at batch 8: with the weight-stationary forward GEMMs (csrc/wsgemm.hip:
# every launch within half a bf16 ulp of an fp64 evaluation, tests/test_wsgemm_gpu.py, but another fp32 summation order) the same step
# gives h 4.9 % away from the tile kernels', loss 1.1345 against 1.1216 and a global gradient norm of 236.3 against 253.6 (-6.8 %,
# tools/ws_grad_diff.py), while the weight-stationary BACKWARD alone (forward bits unchanged) moves the norm by 0.15 %: at batch 8 the
# norm is as chaotic as everything else behind 64 train-mode BatchNorms. Measured against the emulation now: 0.072 / 0.061; bound 1.6x.
# The well-conditioned statement of the gradients is test_b256_gpu.py's, at the timed batch.)


def _hip_step0_gnorm(k, x_i, x_j, tape):
    """global gradient norm (pre-clip) and loss of step 0 on the HIP path with the given neighbour ids forced"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    model = build(k)
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).train()
    opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
    F_.TAPE = F_.KnnTape(replay=tape)
    try:
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i.to(DEV), x_j.to(DEV))
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
    finally:
        F_.TAPE = None
    opt.step()
    torch.cuda.synchronize()
    return float(opt.grad_norm), float(loss.detach())


def test_bf16_train_step_pinned_arithmetic_keeps_the_tight_gradient_norm_bound(bf16_mode, golden):
    """with the weight-stationary FORWARD kernels off (tuning key ws_gemm bit 0: the tile kernels' summation order, which
    the oracle's emulation was calibrated against in rounds 2-4; the weight-stationary backward stays on) the global gradient norm of
    the B = 8 golden step must stay within 0.02 of the emulation's — the bound of rounds 2-4 (measured then: 0.0011 / 0.0026). A real
    gradient bug of a few percent in any kernel other than ws_fwd cannot hide behind the 0.12 that the default path needs."""
    from neuralsampleid_amd import ops
    g = golden("e2e_b8_k3")
    em = emulated_step0(g, 3)
    ops.set_tuning("ws_gemm", 6)
    try:
        ops.launch_counters(reset=True)
        gn, ls = _hip_step0_gnorm(3, g.t("x_i"), g.t("x_j"), tape_of(g, "s0"))
        cnt = ops.launch_counters()
    finally:
        ops.reset_tuning()
    assert cnt["ws_fwd"] == 0 and cnt["ws_bwd_data"] + cnt["ws_bwd_bnapply"] > 0, cnt
    rel = (gn - em["gnorm"]) / em["gnorm"]
    note("bf16_train_pinned_ws6", {"gnorm_rel_signed": rel, "dloss": ls - em["loss"]})
    print("pinned arithmetic (ws_gemm=6): signed gnorm rel", rel, "dloss", ls - em["loss"])
    assert abs(rel) < 0.02 and abs(ls - em["loss"]) < 0.04


def test_weight_stationary_forward_moves_the_gradient_norm_as_signed_noise(bf16_mode, golden):
    """the default path (ws_gemm = 7) sits 6-7 % from the emulation's gradient norm on the golden batch. If that were a
    systematic error of the weight-stationary forward (statistics, affine on load) it would keep its sign under small input perturbations;
    chaos behind 64 train-mode BatchNorms at batch 8 does not. Six perturbed batches (x + 0.5 dB of noise), each against the oracle's
    bf16 emulation of THAT batch with the emulation's own neighbour ids forced on the HIP side: the signed relative differences must show
    both signs and a mean inside 0.09 (measured: -0.093 ... +0.137, mean -0.004)."""
    from oracle import ref_torch as R
    g = golden("e2e_b8_k3")
    with open(os.path.join(GOLDEN, "state_shapes.json")) as f:
        shapes = {k_: tuple(v) for k_, v in json.load(f).items()}
    from synth import synth_tensor
    torch.set_num_threads(8)
    signed = []
    for seed in range(6):
        gen = torch.Generator().manual_seed(9100 + seed)
        x_i = g.t("x_i") + 0.5 * torch.randn(g.t("x_i").shape, generator=gen)
        x_j = g.t("x_j") + 0.5 * torch.randn(g.t("x_j").shape, generator=gen)
        P = {k_: synth_tensor(k_, torch.empty(s_)) for k_, s_ in shapes.items()}
        keys = R.trainable_keys(P)
        for k_ in keys:
            P[k_].requires_grad_(True)
        R.STORAGE, R.TAPE = "bf16", R.KnnTape()
        try:
            _, _, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, R.encoder_plan("t", 3), True, R.BNState())
            R.ntxent(z_i, z_j, GRAFP_CFG["tau"]).backward()
            tape = list(R.TAPE.recorded)
        finally:
            R.STORAGE, R.TAPE = None, None
        gn_em = float(torch.sqrt(sum(P[k_].grad.double().pow(2).sum() for k_ in keys if P[k_].grad is not None)))
        gn, _ = _hip_step0_gnorm(3, x_i, x_j, tape)
        signed.append((gn - gn_em) / gn_em)
    mean = sum(signed) / len(signed)
    note("bf16_train_ws_signed_noise", {"signed_gnorm_rel": signed, "mean": mean})
    print("signed gnorm differences (default path against the emulation):", signed, "mean", mean)
    assert min(signed) < 0 < max(signed), signed
    assert abs(mean) < 0.09, (mean, signed)       # (the six values re-roll from run to run: sigma of the mean ~0.035)
    assert max(abs(v) for v in signed) < 0.3, signed          # (measured: -0.093 ... +0.137, mean -0.004)


BLOCKS = [("c64n256_k3d1", 64, 256, 3, 1), ("c128n128_k5d1", 128, 128, 5, 1), ("c256n64_k18d3", 256, 64, 18, 3),
          ("c512n32_k3d1", 512, 32, 3, 1)]


@pytest.mark.parametrize("tag,C,N,k,d", BLOCKS)
def test_block_train_bf16_vs_emulation(bf16_mode, golden, tag, C, N, k, d):
    """ONE Grapher + FFN block, training mode, bf16 storage: forward, input gradient, parameter gradients and running
    statistics against the oracle's bf16 emulation on the same (bf16-representable) input — the un-amplified statement of
    the timed arithmetic: one block deep, rounding differences are not yet multiplied by the layers behind them."""
    import torch.nn as nn
    from conftest import to_rows
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.gcn_lib.torch_vertex import Grapher
    from neuralsampleid_amd.encoder.graph_encoder import FFN
    from oracle import ref_torch as R
    g = golden("block_" + tag)
    blk = nn.Sequential(Grapher(C, k, d, "mr", "relu", "batch", True, False, 0.2, 1, n=N, drop_path=0.0,
                                relative_pos=True), FFN(C, 4 * C, C, act="relu", drop_path=0.0))
    sd = synth_state(blk.state_dict(), "blk.")
    blk.load_state_dict(sd)
    blk.to(DEV).train()
    x = g.t("x").to(torch.bfloat16).float()                         # what the bf16 pipeline holds at a block boundary
    gout = g.t("gout").to(torch.bfloat16).float()
    P = {k_: v.clone() for k_, v in sd.items()}
    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    xr = to_rows(x).requires_grad_(True)
    R.STORAGE, R.TAPE = "bf16", R.KnnTape()
    try:
        st = R.BNState()
        y_ref = R.ffn(R.grapher(xr, P, "0.", k, d, True, st), P, "1.", True, st)
        y_ref.backward(to_rows(gout))
        idx = R.TAPE.recorded[0]
    finally:
        R.STORAGE, R.TAPE = None, None
    F_.TAPE = F_.KnnTape(replay=[idx])
    xg = x.to(DEV).requires_grad_(True)
    y = blk(xg)
    y.backward(gout.to(DEV))
    F_.TAPE = None
    from conftest import from_rows
    m = {"y": relerr(y, from_rows(y_ref.detach())), "dx": relerr(xg.grad, from_rows(xr.grad))}
    pg = {}
    for name, p in blk.named_parameters():
        ref = P[name].grad
        if ref is None or float(ref.norm()) < 1e-3 * max(1.0, float(ref.numel()) ** 0.5 * 1e-3):
            continue
        pg[name] = relerr(p.grad, ref)
    m["param_grad_worst"], m["param_grad_worst_name"] = max(pg.values()), max(pg, key=pg.get)
    m["param_grad_median"] = sorted(pg.values())[len(pg) // 2]
    m["bn"] = max(maxerr(b.float(), st.updates[n].float()) / max(1.0, float(st.updates[n].abs().max()))
                  for n, b in blk.named_buffers() if n in st.updates and not n.endswith("num_batches_tracked"))
    note(f"block_bf16_{tag}", m)
    print("measured", json.dumps(m))
    assert m["y"] < TOLK["y"] and m["dx"] < TOLK["dx"] and m["bn"] < TOLK["bn"]
    assert m["param_grad_median"] < TOLK["pg_median"] and m["param_grad_worst"] < TOLK["pg_worst"], m


# measured on MI355X in round 2 (three block shapes): y 0.9e-3 .. 2.1e-3 relative L2 (below one bf16 ulp, 3.9e-3), dx 6.4e-3 ..
# 1.0e-2, parameter gradients median 5.6e-3 .. 9.9e-3, worst 0.9e-2 .. 2.0e-2 (a BatchNorm bias), running statistics <= 4.6e-5
TOLK = {"y": 6e-3, "dx": 0.03, "bn": 1.5e-4, "pg_median": 0.03, "pg_worst": 0.06}


def test_bench_configuration_one_replay_vs_eager_steps(bf16_mode):
    """The EXACT configuration bench.py times — seed-42 default init, B = 256 synthetic clip pairs, bf16 storage,
    SimCLR(overlap_views=True) (two HIP streams, BwdChain cross-block links, deferred running statistics),
    FusedClipAdam(direct_grads), the whole step in ONE hipGraph — replayed once from the initial state, against
      (1) one EAGER, single-stream step in the same bf16 arithmetic: the combination itself (orchestration only: atomics
          order is the only difference), and
      (2) one eager, single-stream, strict-fp32 HIP step (the parity path of DESIGN.md section 4, itself checked against
          the reference goldens in tests/test_e2e_gpu.py): what the bf16 arithmetic costs at the timed size.
    All three runs use the neighbour indices of the fp32 run (kNN is discontinuous)."""
    import bench
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops, parallel
    from neuralsampleid_amd.graphs import GraphedTrainStep
    from neuralsampleid_amd.optim import FusedClipAdam
    B = 256
    x_i, x_j = bench.synth_clips(B, 1000, DEV)

    def fresh(overlap):
        torch.manual_seed(42)
        model = build(3, overlap).to(DEV).train()
        return model, FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)

    def eager(mode, tape):
        ops.set_gemm_precision(mode)
        F_.set_activation_dtype(mode)
        model, opt = fresh(False)
        p0 = opt.flat_p.clone()
        F_.TAPE = F_.KnnTape(replay=tape)
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i, x_j)
        loss = parallel.dist_ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        rec = [t.clone() for t in F_.TAPE.recorded]
        F_.TAPE = None
        opt.step()
        torch.cuda.synchronize()
        return {"loss": float(loss.detach()), "g": opt.flat_g.clone(), "dp": opt.flat_p - p0, "gn": float(opt.grad_norm),
                "bn": {n: b.clone() for n, b in model.named_buffers()}, "p0": p0, "tape": rec}

    ref32 = eager("fp32", None)
    tape = ref32["tape"]
    assert len(tape) == 24
    ref16 = eager("bf16", tape)

    # ---- the timed configuration
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    model, opt = fresh(True)
    p0 = ref32["p0"]
    assert torch.equal(opt.flat_p, p0)
    F_.TAPE = F_.KnnTape(replay=tape, cyclic=True)          # device int32 tensors: the captured kernels read them in place
    step = GraphedTrainStep(model, opt, GRAFP_CFG, x_i, x_j)
    F_.TAPE = None
    assert torch.equal(opt.flat_p, p0) and int(opt.step_count) == 0        # construction left the training state alone
    assert all(int(b) == 0 for n, b in model.named_buffers() if n.endswith("num_batches_tracked"))
    loss = step(x_i, x_j)
    torch.cuda.synchronize()
    assert int(opt.step_count) == 1 and model._side_stream is not None
    for n, b in model.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == 2, n                           # both views, view i first (simclr.py:36,42)
    lr = GRAFP_CFG["lr"]
    dp = opt.flat_p - p0

    def compare(ref):
        return {"dloss": abs(float(loss.detach()) - ref["loss"]), "rel_flat_g": relerr(opt.flat_g, ref["g"]),
                "gn_rel": abs(float(opt.grad_norm) - ref["gn"]) / ref["gn"],
                "update_mean_abs_diff_over_lr": float((dp - ref["dp"]).abs().mean()) / lr,
                "bn_worst": max(maxerr(b.double(), ref["bn"][n].double()) / max(1.0, float(ref["bn"][n].double().abs().max()))
                                for n, b in model.named_buffers())}
    m16, m32 = compare(ref16), compare(ref32)
    m32["loss_ref"] = ref32["loss"]
    note("bench_config_replay", {"vs_eager_bf16": m16, "vs_eager_fp32": m32})
    print("measured", json.dumps({"vs_eager_bf16": m16, "vs_eager_fp32": m32}, indent=1))
    # (1) same arithmetic, different orchestration
    assert m16["dloss"] < TOLB["dloss16"] and m16["rel_flat_g"] < TOLB["g16"] and m16["gn_rel"] < TOLB["gn16"]
    assert m16["update_mean_abs_diff_over_lr"] < TOLB["upd16"] and m16["bn_worst"] < TOLB["bn16"]
    # (2) bf16 against strict fp32 at the timed size
    assert m32["dloss"] < TOLB["dloss32"] and m32["rel_flat_g"] < TOLB["g32"] and m32["gn_rel"] < TOLB["gn32"]
    assert m32["update_mean_abs_diff_over_lr"] < TOLB["upd32"] and m32["bn_worst"] < TOLB["bn32"]


# measured on MI355X in round 2 — (1) replay vs eager bf16: loss and running statistics bit-identical (0.0), flat gradient
# 1.3e-2 relative L2 (fp32 atomics order -> rare one-ulp flips of bf16-stored gradients, amplified by 60 layers of backward),
# global norm 0.24 %, first Adam update 0.003 lr apart on average; (2) vs strict fp32: |dloss| 0.042 on a loss of 4.65,
# flat gradient 0.60 relative L2 (gradient ROUTING through max / ReLU choices, see TOL above), global norm 4.2 %, update
# 0.24 lr, running statistics 0.28 %.
# gn16 bounds a NOISE variable (the difference of two global norms whose gradients differ by atomics order): over 20 runs in round 3
# it ranged 0.0002 ... 0.0052 (once past the former bound of 0.008 in about ten full-suite runs) while rel_flat_g stayed at
# 0.012-0.017; the bound is now 4x the largest value seen
TOLB = {"dloss16": 1e-3, "g16": 0.04, "gn16": 0.02, "upd16": 0.01, "bn16": 1e-4,
        "dloss32": 0.12, "g32": 1.2, "gn32": 0.2, "upd32": 0.6, "bn32": 0.008}      # gn32: 0.042 / 0.055 / 0.059 seen (same kind of variable)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_eval_after_fused_and_graphed_training_sees_current_state(golden, mode):
    """train -> extract_fingerprints -> train (eager FusedClipAdam steps, then hipGraph replays) -> extract_fingerprints:
    the kernels write weights and running statistics behind torch's version counters, so every eval-side cache (bf16
    weight shadows, eval-mode BatchNorm affines, folded conv+BN weights) must follow ops.WEIGHT_EPOCH / STATS_EPOCH.
    Each eval must equal the eval of a FRESH model loaded from the trained model's state_dict (same kernels, same numbers;
    the projector's split-K GEMM adds with fp32 atomics, so equality is to 2e-6, while a stale cache is off by > 1e-3), and in
    fp32 the oracle's eval forward of that state_dict."""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.fingerprint import extract_fingerprints
    from neuralsampleid_amd.graphs import GraphedTrainStep
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from oracle import ref_torch as R
    g = golden("e2e_b8_k3")
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    ops.set_gemm_precision(mode)
    F_.set_activation_dtype(mode)
    try:
        model = build(3)
        model.load_state_dict(synth_state(model.state_dict()))
        model.to(DEV).train()
        opt = FusedClipAdam(model.parameters(), lr=1e-3, max_norm=1.0)      # a visible step size

        def fresh_eval():
            twin = build(3)
            twin.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
            return extract_fingerprints(twin.to(DEV), x_i, batch=8)

        def eager_steps(n):
            for _ in range(n):
                opt.zero_grad()
                _, _, z_i, z_j = model(x_i, x_j)
                ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
                opt.step()

        z0 = extract_fingerprints(model, x_i, batch=8)
        assert maxerr(z0, fresh_eval()) < 2e-6
        eager_steps(2)
        z1 = extract_fingerprints(model, x_i, batch=8)
        assert maxerr(z1, z0) > 1e-3                                        # the weights really moved
        assert maxerr(z1, fresh_eval()) < 2e-6, maxerr(z1, fresh_eval())    # stale caches would give z0-like values
        step = GraphedTrainStep(model, opt, GRAFP_CFG, x_i, x_j, loss_fn=ntxent_loss)
        assert maxerr(extract_fingerprints(model, x_i, batch=8), z1) < 2e-6  # construction restored the training state
        step(x_i, x_j)
        step(x_j, x_i)
        z2 = extract_fingerprints(model, x_i, batch=8)
        assert maxerr(z2, z1) > 1e-3
        assert maxerr(z2, fresh_eval()) < 2e-6, maxerr(z2, fresh_eval())
        if mode == "fp32":                                                  # and the oracle agrees on what z2 should be
            P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            with torch.no_grad():
                _, _, z_ref, _ = R.simclr_forward(g.t("x_i"), g.t("x_i"), P, GRAFP_CFG, R.encoder_plan("t", 3), False, None)
            cos = torch.nn.functional.cosine_similarity(z2.cpu(), z_ref, dim=1)
            assert float(cos.min()) > 0.9995 and float(cos.median()) > 0.99999     # own kNN: near-tie flips only
    finally:
        ops.set_gemm_precision("fp32")
        F_.set_activation_dtype("fp32")


def test_weight_shadows_are_not_reconverted_every_step(bf16_mode):
    """FusedClipAdam refreshes all bf16 weight shadows with ONE launch per step; the per-weight lookups of the GEMM wrappers
    must then find them fresh (a per-weight conversion launch per GEMM cost 0.3 ms per step when a version-counter mix-up
    made every lookup miss: 68 conversions per step instead of 7 in the rocprofv3 trace)."""
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    model = build(3)
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).train()
    opt = FusedClipAdam(model.parameters(), lr=1e-4)
    x = torch.randn(4, 64, 128, device=DEV)
    calls = []
    real = ops.f32_to_bf16
    ops.f32_to_bf16 = lambda src, dst=None: (calls.append(src.numel()), real(src, dst))[1]
    try:
        for _ in range(3):
            calls.clear()
            opt.zero_grad()
            _, _, z_i, z_j = model(x, x + 0.1)
            ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
            opt.step()
    finally:
        ops.f32_to_bf16 = real
    # one flat refresh + the three downsample layers' packed weights (forward: (Cout, 3C); backward: [W_2 ; W_0]) x 2 views
    assert len(calls) <= 1 + 3 * 2 * 2 and max(calls) == opt.numel, calls
