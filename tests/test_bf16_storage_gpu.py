"""GPU parity of bf16 ACTIVATION STORAGE (functional.set_activation_dtype('bf16'); BASELINE config 2: bf16 storage, fp32
accumulate). Every kernel that touches an activation tensor is run on bf16 tensors and compared with the same op on the
upcast fp32 copy of the SAME bf16 values (so the only differences are the final rounding of the outputs to bf16:
relative L2 <= 2.5e-3, i.e. bf16's 2^-9 half-ulp rms, and bit-exact for pure data movement / integer outputs)."""
import numpy as np
import pytest

import torch

from compare import relerr
from synth import GRAFP_CFG, synth_randn, synth_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture()
def ops():
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops as o
    o.set_gemm_precision("bf16")
    yield o
    o.reset_tuning()
    o.set_gemm_precision("fp32")
    F_.set_activation_dtype("fp32")


def act_ref(x, act):
    return {0: x, 1: torch.relu(x), 2: torch.nn.functional.leaky_relu(x, 0.2)}[act]


def bfr(x):
    """fp64 view of the bf16 rounding of x"""
    return x.to(BF).double()


CASES = [(512, 64, 64, 1, False, 0), (200, 256, 64, 1, True, 1), (384, 32, 32, 4, False, 0), (256, 128, 128, 4, True, 2),
         (640, 64, 8, 1, False, 0), (2048, 1024, 256, 1, True, 1), (136, 192, 384, 1, False, 0),
         (512, 2048, 512, 1, True, 1), (256, 1024, 1024, 1, False, 0),   # >= 64 tiles of 128x128: the 8-wave wgrad form
         (16384, 256, 512, 1, True, 1), (16384, 1024, 256, 1, False, 0),  # 128x64 weight-gradient tiles (>= 256 workgroups of 1024 rows)
         (16384, 1024, 256, 1, True, 1), (16384, 256, 1024, 1, True, 1),  # >= 1024 tiles of 128x128: 8-wave 256x128 forward (affine
         (32768, 512, 128, 1, False, 0)]                                  # + statistics: two 128-row tiles per workgroup) / backward-data


@pytest.mark.parametrize("M,Nout,K,groups,affine,act", CASES)
def test_linear_family_bf16_storage(ops, M, Nout, K, groups, affine, act):
    x = synth_randn(f"sx{M}{K}{groups}", M, groups * K).to(BF)
    w = synth_randn(f"sw{Nout}{K}{groups}", groups * Nout, K) * K ** -0.5
    bias = synth_randn(f"sb{Nout}", groups * Nout)
    dout = synth_randn(f"sd{M}{Nout}", M, groups * Nout).to(BF)
    add = synth_randn(f"sa{M}{K}", M, groups * K).to(BF)
    sc = 1 + 0.2 * synth_randn("ssc", groups * K) if affine else None
    sh = 0.3 * synth_randn("ssh", groups * K) if affine else None
    xin = act_ref(x.float() * sc + sh, act) if affine else x.float()      # the kernel's fp32 value before bf16 rounding
    d = lambda t: None if t is None else t.to(DEV)
    G = range(groups)
    ref = torch.cat([bfr(xin[:, g * K:(g + 1) * K]) @ bfr(w[g * Nout:(g + 1) * Nout]).t() for g in G], 1) + bias.double()
    out, stat = ops.linear_fwd(d(x), d(w), d(bias), M, Nout, K, groups, d(sc), d(sh), act, 0, want_stat=True)
    assert out.dtype == BF and stat.dtype == torch.float32
    assert relerr(out, ref) < 2.5e-3
    assert relerr(stat[0].sum(0), ref.sum(0)) < 1e-4                      # statistics come from the fp32 accumulators
    assert relerr(stat[1].sum(0), (ref * ref).sum(0)) < 1e-4
    ref = torch.cat([dout.double()[:, g * Nout:(g + 1) * Nout] @ bfr(w[g * Nout:(g + 1) * Nout]) for g in G], 1)
    got = ops.linear_bwd_data(d(dout), d(w), M, Nout, K, groups, d(add))
    assert got.dtype == BF and relerr(got, ref + add.double()) < 2.5e-3
    ref = torch.cat([dout.double()[:, g * Nout:(g + 1) * Nout].t() @ bfr(xin[:, g * K:(g + 1) * K]) for g in G], 0)
    dw = torch.zeros(groups * Nout, K, device=DEV)
    ops.linear_bwd_weight(d(dout), d(x), dw, M, Nout, K, groups, d(sc), d(sh), act)
    assert relerr(dw, ref) < 2e-5                                          # fp32 output: only accumulation order differs


@pytest.mark.parametrize("M,C,act", [(512, 64, 0), (300, 256, 1), (1024, 2048, 1), (256, 80, 2)])
def test_batchnorm_kernels_bf16_storage(ops, M, C, act):
    r16 = (synth_randn(f"sbn{M}{C}", M, C) * 1.5 + 0.7).to(BF).to(DEV)
    res16 = synth_randn("sbnres", M, C).to(BF).to(DEV)
    dout16 = synth_randn("sbndo", M, C).to(BF).to(DEV)
    r32, res32, dout32 = r16.float(), res16.float(), dout16.float()
    x64 = r32.double()
    mean, var = x64.mean(0), x64.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    gamma = (1 + 0.1 * synth_randn("sbng", C)).to(DEV)
    beta = (0.1 * synth_randn("sbnb", C)).to(DEV)
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - mean * gamma.double() * invstd).float()
    aff = ops.BNAffine(scale.contiguous(), shift.contiguous(), mean.float().contiguous(), invstd.float().contiguous())
    y16 = ops.bn_apply(r16, aff, act, res16)
    y32 = ops.bn_apply(r32, aff, act, res32)
    assert y16.dtype == BF and relerr(y16, y32) < 2.5e-3
    g16, b16 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    g32, b32 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dr16 = ops.bn_backward(dout16, r16, aff, act, g16, b16)
    dr32 = ops.bn_backward(dout32, r32, aff, act, g32, b32)
    assert dr16.dtype == BF and relerr(dr16, dr32) < 2.5e-3
    assert relerr(g16, g32) < 1e-5 and relerr(b16, b32) < 1e-5           # reductions are fp32 on identical inputs
    s16, s32 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.colsum_acc(dout16, s16)
    ops.colsum_acc(dout32, s32)
    assert relerr(s16, s32) < 1e-5


def test_graph_kernels_bf16_storage(ops):
    B, N, C, k = 3, 128, 128, 5
    r16 = synth_randn("sgr", B * N, C).to(BF).to(DEV)
    sc, sh = (1 + 0.3 * synth_randn("sgsc", C)).to(DEV), (0.5 * synth_randn("sgsh", C)).to(DEV)
    aff = ops.BNAffine(sc, sh)
    idx16 = ops.knn_graph(r16, B, N, C, k, 2, aff)
    idx32 = ops.knn_graph(r16.float(), B, N, C, k, 2, aff)
    assert torch.equal(idx16, idx32)                                       # same values in -> same graph out
    u16, am16 = ops.mr_aggregate_fwd(r16, idx16, B, N, C, aff)
    u32, am32 = ops.mr_aggregate_fwd(r16.float(), idx16, B, N, C, aff)
    assert u16.dtype == BF and torch.equal(am16, am32) and torch.equal(u16, u32.to(BF))
    du16 = synth_randn("sgdu", B * N, 2 * C).to(BF).to(DEV)
    dy16 = ops.mr_aggregate_bwd(du16, idx16, am16, B, N, C)
    dy32 = ops.mr_aggregate_bwd(du16.float(), idx16, am16, B, N, C)
    assert dy16.dtype == BF and relerr(dy16, dy32) < 2.5e-3


def test_misc_kernels_bf16_storage(ops):
    B, N, C = 3, 64, 64
    x16 = synth_randn("smx", B * N, C).to(BF).to(DEV)
    col16 = ops.im2col3_fwd(x16, B, N, C)
    assert col16.dtype == BF and torch.equal(col16, ops.im2col3_fwd(x16.float(), B, N, C).to(BF))
    dcol16 = synth_randn("smd", B * ops.ds_out_nodes(N), 3 * C).to(BF).to(DEV)
    assert relerr(ops.im2col3_bwd(dcol16, B, N, C), ops.im2col3_bwd(dcol16.float(), B, N, C)) < 2.5e-3
    assert relerr(ops.node_mean_fwd(x16, B, N, C), ops.node_mean_fwd(x16.float(), B, N, C)) < 1e-6
    dm = synth_randn("smm", B, C).to(DEV)
    assert torch.equal(ops.node_mean_bwd(dm, B, N, C, BF), ops.node_mean_bwd(dm, B, N, C).to(BF))
    x4 = synth_randn("sml", 2, 40, 24).to(DEV)
    rows16 = ops.bcn_to_rows(x4, BF)
    assert rows16.dtype == BF and torch.equal(rows16, ops.bcn_to_rows(x4).to(BF))
    assert torch.equal(ops.rows_to_bcn(rows16, 2, 24), rows16.float().reshape(2, 24, 40).transpose(1, 2))
    spec = (synth_randn("sms", 4, 64, 128) * 20 - 40).to(DEV)
    w, b = synth_randn("smw", 8, 3, 4, 8).to(DEV) * 0.2, synth_randn("smb", 8).to(DEV) * 0.1
    o16, mm = ops.peak_patchify_fwd(spec, w, b, 4, 8, BF)
    o32, _ = ops.peak_patchify_fwd(spec, w, b, 4, 8)
    assert o16.dtype == BF and torch.equal(o16, o32.to(BF))
    do16 = synth_randn("smdo", 4 * 256, 8).to(BF).to(DEV)
    dw16, db16 = torch.zeros_like(w), torch.zeros_like(b)
    dw32, db32 = torch.zeros_like(w), torch.zeros_like(b)
    ops.peak_patchify_bwd(spec, mm, o16, do16, 4, 8, dw16, db16)
    ops.peak_patchify_bwd(spec, mm, o16.float(), do16.float(), 4, 8, dw32, db32)
    assert relerr(dw16, dw32) < 1e-4 and relerr(db16, db32) < 1e-4


def build(k=3):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=8, k=k, size="t"))


def test_simclr_eval_bf16_storage_vs_reference(ops, golden):
    """fingerprinting semantics with bf16 activations end to end, against the fp32 REFERENCE goldens"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    F_.set_activation_dtype("bf16")
    g = golden("e2e_b8_k3")
    model = build()
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).eval()
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    n = len([k for k in g if k.startswith("knn.eval.")])
    gold_idx = [g.t(f"knn.eval.{c}") for c in range(n)]
    try:
        F_.TAPE = F_.KnnTape(replay=gold_idx)
        with torch.no_grad():
            h_i, h_j, z_i, z_j = model(x_i, x_j)
            loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    finally:
        F_.TAPE = None
    assert h_i.dtype == torch.float32 and z_i.dtype == torch.float32
    print("bf16 storage eval: rel_h", relerr(h_i, g.t("h_i_eval")), "dloss", abs(float(loss.detach()) - float(g["loss_eval"][0])))
    assert relerr(h_i, g.t("h_i_eval")) < 4e-2 and relerr(h_j, g.t("h_j_eval")) < 4e-2
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 5e-3
    cos = torch.nn.functional.cosine_similarity(z_i.cpu(), g.t("z_i_eval"), dim=1)
    assert float(cos.min()) > 0.999


def test_training_curve_bf16_storage_tracks_fp32(ops):
    """same acceptance as tests/test_bf16_gpu.py::test_training_curve_bf16_tracks_fp32, with bf16 activation storage"""
    import math
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    B = 128
    gi, gj = torch.Generator().manual_seed(0), torch.Generator().manual_seed(1)
    x_i = torch.randn(B, 64, 128, generator=gi) * 20 - 40
    x_j = (x_i + 3 * torch.randn(B, 64, 128, generator=gj)).to(DEV)
    x_i = x_i.to(DEV)
    curves = {}
    for mode in ("fp32", "bf16"):
        ops.set_gemm_precision(mode)
        F_.set_activation_dtype(mode)
        torch.manual_seed(42)
        model = build().to(DEV).train()
        opt = FusedClipAdam(model.parameters(), lr=8e-5, max_norm=1.0)
        ls = []
        for _ in range(30):
            opt.zero_grad()
            _, _, z_i, z_j = model(x_i, x_j)
            loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        curves[mode] = ls
    ratios = [abs(math.log(a / b)) for a, b in zip(curves["bf16"], curves["fp32"])]
    worst, mean = max(ratios), sum(ratios) / len(ratios)
    print("bf16-storage curve:", [round(v, 3) for v in curves["bf16"][::5]], "fp32:", [round(v, 3) for v in curves["fp32"][::5]],
          "worst", round(worst, 3), "mean", round(mean, 3))
    # measured at B=128 over repeated runs: mean 0.29-0.52, worst 1.1-1.3 (the worst sits in the tail where both losses are
    # ~0.02-0.05; a re-seeded fp32 run differs from another fp32 run by about as much: kNN near-ties make training chaotic)
    assert mean < 0.8 and worst < 2.0, (mean, worst, curves)
    assert curves["bf16"][-1] < 0.1 * curves["bf16"][0]


@pytest.mark.parametrize("M,Nout,K,groups,act,add", [(512, 256, 1024, 1, 1, False), (640, 64, 128, 1, 1, True),
                                                       (384, 128, 64, 1, 0, False), (200, 64, 256, 1, 2, True),
                                                       (256, 32, 32, 4, 1, False),
                                                       (16384, 256, 1024, 1, 1, True),      # 8-wave 256x128 tiles: two partial rows each
                                                       (16384, 128, 1024, 1, 0, False)])
def test_bwd_data_emits_bn_backward_sums(ops, M, Nout, K, groups, act, add):
    """linear_bwd_data(bn=...) = the plain GEMM plus the column sums nsid_bn_bwd_reduce computes from its stored output"""
    C = groups * K
    dout = synth_randn(f"fd{M}{Nout}", M, groups * Nout).to(BF).to(DEV)
    w = (synth_randn(f"fw{Nout}{K}", groups * Nout, K) * Nout ** -0.5).to(DEV)
    r = (synth_randn(f"fr{M}{C}", M, C) * 1.3 + 0.2).to(BF).to(DEV)
    addend = synth_randn(f"fa{M}{C}", M, C).to(BF).to(DEV) if add else None
    mean, var = r.float().mean(0), r.float().var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    gamma, beta = (1 + 0.2 * synth_randn("fg", C)).to(DEV), (0.1 * synth_randn("fb", C)).to(DEV)
    aff = ops.BNAffine(gamma * invstd, beta - mean * gamma * invstd, mean, invstd)
    plain = ops.linear_bwd_data(dout, w, M, Nout, K, groups, addend)
    fused, partial = ops.linear_bwd_data(dout, w, M, Nout, K, groups, addend, bn=(r, aff, act))
    assert torch.equal(fused, plain) and partial.shape == (2, ops.row_tiles(M), C)
    dg0, db0, dg1, db1 = (torch.zeros(C, device=DEV) for _ in range(4))
    ref = ops.bn_backward(plain, r, aff, act, dg0, db0)
    got = ops.bn_backward(fused, r, aff, act, dg1, db1, partial=partial)
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5          # same inputs, different summation order
    assert relerr(got.float(), ref.float()) < 1e-5



@pytest.mark.parametrize("M,Nout,K,groups,act,add,link,fused_expected", [
    (16384, 1024, 256, 1, 1, True, True, True),     # FFN hidden layer at C = 256: two column tiles, side output from tile 0
    (8192, 2048, 512, 1, 1, True, False, True),     # C = 512: four column tiles
    (65536, 256, 64, 1, 1, True, True, True),       # C = 64: 64-wide tiles
    (16384, 128, 128, 4, 1, False, False, True),    # grouped layer at C = 256
    (32768, 128, 128, 1, 0, True, True, True),      # Grapher fc1 (no activation) at C = 128: one 128-wide column tile
    (2048, 64, 64, 1, 0, True, True, True),         # the batch-8 goldens' stage 0
    (2048, 32, 32, 4, 1, False, False, False),      # grouped layer at C = 64: outside the fused form -> two-call fallback
    (200, 64, 256, 1, 2, True, False, False)])      # ragged rows: fallback
def test_bn_backward_on_the_backward_data_operand_load(ops, M, Nout, K, groups, act, add, link, fused_expected):
    """ops.bn_backward_linear_bwd_data: BatchNorm(+act) backward evaluated on the backward-data GEMM's operand load (csrc/gemm.hip
    ABN) against the two-call form (bn_bwd_apply pass + plain GEMM): same dr up to one fp32 rounding before the bf16 store, same din,
    same column sums for the next BatchNorm backward, same dgamma / dbeta; the launch counter says which form ran.
    (The TILE kernels' form: the weight-stationary one of csrc/wsgemm.hip, which takes these shapes by default where it has them, is
    held to the same statement by tests/test_wsgemm_gpu.py; ws_gemm = 3 keeps it off the operand-load path here.)"""
    from neuralsampleid_amd._lib import launch_counters
    ops.set_tuning("ws_gemm", 3)
    C = groups * Nout
    dy = synth_randn(f"ady{M}{C}", M, C).to(BF).to(DEV)
    r = (synth_randn(f"ar{M}{C}", M, C) * 1.3 + 0.2).to(BF).to(DEV)
    w = (synth_randn(f"aw{Nout}{K}{groups}", groups * Nout, K) * Nout ** -0.5).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    addend = synth_randn(f"aa{M}{K}", M, groups * K).to(BF).to(DEV) if add else None
    mean, var = r.float().mean(0), r.float().var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    gamma, beta = (1 + 0.2 * synth_randn("ag", C)).to(DEV), (0.1 * synth_randn("ab", C)).to(DEV)
    aff = ops.BNAffine(gamma * invstd, beta - mean * gamma * invstd, mean, invstd)
    bn = False
    if link:           # din is dL/dy of an upstream BatchNorm layer with raw input r_up
        CK = groups * K
        r_up = (synth_randn(f"au{M}{CK}", M, CK) * 0.8 - 0.1).to(BF).to(DEV)
        mu, vu = r_up.float().mean(0), r_up.float().var(0, unbiased=False)
        iu = 1 / torch.sqrt(vu + 1e-5)
        bn = (r_up, ops.BNAffine(iu, -mu * iu, mu, iu), 1)
    out = {}
    for fuse in (False, True):
        ops.FUSE_BN_BWD_APPLY = fuse
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        launch_counters(reset=True)
        try:
            dr, din, part = ops.bn_backward_linear_bwd_data(dy.clone(), r, aff, act, dg, db, None, w, M, Nout, K, groups,
                                                            addend=addend, bn=bn)
        finally:
            ops.FUSE_BN_BWD_APPLY = True
        torch.cuda.synchronize()
        out[fuse] = (dr.float(), din.float(), part, dg, db, launch_counters())
    (dr0, din0, p0, dg0, db0, c0), (dr1, din1, p1, dg1, db1, c1) = out[False], out[True]
    assert c0["gemm_bn_apply_load"] == 0 and c0["bn_bwd_apply"] == 1
    assert c1["gemm_bn_apply_load"] == (1 if fused_expected else 0) and c1["bn_bwd_apply"] == (0 if fused_expected else 1), c1
    assert relerr(dg1, dg0) < 1e-6 and relerr(db1, db0) < 1e-6                    # same reduce + finalize
    # dr: the same fp32 value up to the association of three terms, rounded to bf16 once: rare one-ulp (2^-8 relative) differences
    assert relerr(dr1, dr0) < 1e-3, relerr(dr1, dr0)
    assert float(((dr1 - dr0).abs() > 2 ** -7 * dr0.abs().clamp_min(1e-3)).float().mean()) < 1e-4
    assert relerr(din1, din0) < 2.5e-3, relerr(din1, din0)                        # the GEMM sums those dr values; bf16 store
    if link:
        assert p1.shape == p0.shape and relerr(p1, p0) < 5e-3
    else:
        assert p0 is None and p1 is None
    # against an fp64 evaluation of the textbook formula on the same bf16 inputs
    x, d = r.double(), dy.double()
    z = aff.scale.double() * x + aff.shift.double()
    g = d * {0: torch.ones_like(z), 1: (z > 0).double(), 2: torch.where(z > 0, 1.0, 0.2)}[act]
    xh = (x - mean.double()) * invstd.double()
    dr_ref = aff.scale.double() * (g - g.mean(0) - xh * (g * xh).mean(0))
    assert relerr(dr1, dr_ref) < 3e-3                                             # one bf16 rounding
    G_ = range(groups)
    din_ref = torch.cat([bfr(dr_ref[:, q * Nout:(q + 1) * Nout]) @ bfr(w[q * Nout:(q + 1) * Nout]) for q in G_], 1)
    if add:
        din_ref = din_ref + addend.double()
    assert relerr(din1, din_ref) < 6e-3, relerr(din1, din_ref)


@pytest.mark.parametrize("B,C,k", [(5, 64, 3), (3, 128, 5), (4, 256, 3), (2, 256, 18), (64, 64, 3), (301, 128, 3), (1030, 256, 5)])
def test_eval_mrconv_in_one_launch(ops, B, C, k):
    """csrc/mrconv_fused.hip: max-relative aggregation + grouped conv (+ folded BatchNorm + ReLU) in one launch per clip, the
    interleaved (B*N, 2C) tensor never formed: against the two-launch form (nsid_mr_aggregate_fwd + grouped nsid_linear_fwd with the
    ReLU epilogue: same bf16 rounding of the aggregated values, same MFMA) and against an fp64 evaluation"""
    from neuralsampleid_amd._lib import launch_counters
    N = 16384 // C
    y = synth_randn(f"mcy{B}{C}", B * N, C).to(BF).to(DEV)
    g = torch.Generator().manual_seed(C + k)
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32)
    idx[:, :, 0] = torch.arange(N, dtype=torch.int32)                      # self first, as the graph builder emits it
    idx = idx.to(DEV)
    w = (synth_randn(f"mcw{C}", 2 * C, C // 2) * (C // 2) ** -0.5).to(DEV)
    b = (0.2 * synth_randn(f"mcb{C}", 2 * C)).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    u, _ = ops.mr_aggregate_fwd(y, idx, B, N, C, None, want_argmax=False)
    two, _ = ops.linear_fwd(u, w, b, B * N, C // 2, C // 2, groups=4, act_out=ops.ACT_RELU)
    try:
        # 4 / 8 waves x staged / direct stores, and (bit 2, round 6) one group per workgroup over a range of clips (k <= 8; else the per-clip form)
        for variant in (0, 1, 2, 3, 7, 15):
            ops.set_tuning("mrconv_variant", variant)
            launch_counters(reset=True)
            out = ops.mrconv_fused_fwd(y, idx, B, N, C, w, b)
            torch.cuda.synchronize()
            assert out is not None and launch_counters()["mrconv_fused"] == 1
            assert relerr(out, two) < 1e-3 and float((out != two).float().mean()) < 0.02, variant
    finally:
        ops.reset_tuning()
    yd = y.double().reshape(B, N, C).cpu()
    nbr = torch.gather(yd.unsqueeze(2).expand(B, N, k, C), 1, idx.cpu().long().unsqueeze(-1).expand(B, N, k, C))
    m = (nbr - yd.unsqueeze(2)).max(2).values
    ud = torch.stack([yd, bfr(m.float())], -1).reshape(B * N, 2 * C)        # interleave, aggregated half rounded to bf16
    K = C // 2
    ref = torch.cat([ud[:, q * K:(q + 1) * K] @ bfr(w[q * K:(q + 1) * K]).cpu().t() for q in range(4)], 1) + b.double().cpu()
    assert relerr(out, ref.clamp_min(0)) < 2.5e-3
    assert ops.mrconv_fused_fwd(y[:, :32].contiguous(), idx, B, N, 32, w[:64, :16].contiguous(), b[:64]) is None


@pytest.mark.parametrize("M,C", [(2048, 64), (128, 64), (4096, 128), (65536, 64), (256, 256), (16384, 256), (76800, 256),
                                 (131072, 256), (153600, 64), (76800, 128), (384, 128)])
def test_eval_ffn_in_one_launch(ops, M, C):
    """csrc/ffn256_fused.hip (x tile in registers, weights by LDS-DMA: C = 256, and C = 128 / 64 at whole 256- / 512-row tiles -- 300
    tiles on one workgroup per CU exercise the persistent form's second tile) and csrc/ffn_fused.hip (x tile in LDS: the other rows):
    out = x + W2 relu(W1 x + b1) + b2 (FFN.forward in eval mode with both BatchNorms folded) in one launch,
    the M x 4C hidden tensor never written: against an fp64 evaluation with the same two bf16 rounding points (hidden, output) and
    against the two-launch form (nsid_linear_fwd with the ReLU epilogue + nsid_linear_fwd_res)"""
    from neuralsampleid_amd._lib import launch_counters
    H = 4 * C
    x = synth_randn(f"ffx{M}{C}", M, C).to(BF).to(DEV)
    w1 = (synth_randn(f"ffw1{C}", H, C) * C ** -0.5).to(DEV)
    w2 = (synth_randn(f"ffw2{C}", C, H) * H ** -0.5).to(DEV)
    b1, b2 = (0.3 * synth_randn(f"ffb1{C}", H)).to(DEV), (0.3 * synth_randn(f"ffb2{C}", C)).to(DEV)
    for w in (w1, w2):
        ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    launch_counters(reset=True)
    out = ops.ffn_fused_fwd(x, w1, b1, w2, b2, M, C, H)
    torch.cuda.synchronize()
    assert out is not None and launch_counters()["ffn_fused"] == 1
    hid = torch.relu(x.double() @ bfr(w1).t() + b1.double())
    ref = x.double() + bfr(hid.float()) @ bfr(w2).t() + b2.double()
    assert relerr(out, ref) < 2.5e-3, relerr(out, ref)                       # one bf16 rounding of the output
    err = (out.double() - ref).abs() / (ref.abs() + 1.0)
    assert float(err.max()) < 1.2e-2                                        # no stray element (a mis-indexed tile would be O(1))
    h2, _ = ops.linear_fwd(x, w1, b1, M, H, C, act_out=ops.ACT_RELU)
    two, _ = ops.linear_fwd(h2, w2, b2, M, C, H, addend=x)
    assert relerr(out, two) < 1e-3 and float((out != two).float().mean()) < 0.02      # same rounding points, another summation order
    assert ops.ffn_fused_fwd(x[:, :32].contiguous(), w1[:128, :32].contiguous(), b1[:128], w2[:32, :128].contiguous(), b2[:32],
                             M, 32, 128) is None                            # outside the fused form: the caller falls back


@pytest.mark.parametrize("N,C,k,case", [(256, 64, 18, "random"), (128, 128, 18, "random"), (64, 256, 18, "ties"), (32, 512, 18, "ties"),
                                         (256, 64, 18, "collapse"), (128, 128, 9, "special"), (64, 256, 18, "noaffine")])
def test_integer_key_max_relative_equals_the_scalar_search(ops, N, C, k, case):
    """csrc/mr.hip mr_fwd_key_kernel (bf16 storage, k >= tuning key mr_key_min_k: the deep configuration's k = 18): the neighbour search
    on order-preserving integer keys with the position in the low bits must give mr_fwd_lds_kernel's output and arg-max BYTE FOR BYTE --
    torch.max's first maximum of d = (sc v + sh) - y (gcn_lib/torch_vertex.py:21-32) -- on random data with negative BatchNorm
    scales, on a tie-heavy input (values from a set of five), where different values collapse to the same d (tiny values beside a
    large shift, a zero scale: the wave falls back to the scalar search), and with NaN / +-inf / -0 entries"""
    from neuralsampleid_amd._lib import launch_counters
    B = 64
    g = torch.Generator().manual_seed(N * 7 + k)
    r = torch.randn(B * N, C, generator=g)
    sc = 1 + 0.8 * torch.randn(C, generator=g)            # a fifth of the channels negative
    sh = 0.5 * torch.randn(C, generator=g)
    if case == "ties":
        r = torch.tensor([-1.5, -0.25, 0.0, 0.25, 2.0])[torch.randint(0, 5, (B * N, C), generator=g)]
    elif case == "collapse":
        r = r * 1e-6                                       # 2^-8 steps of 1e-6 vanish beside shifts of 0.5: many neighbours share the maximal d
        sc[::7] = 0.0                                      # a zero scale: every neighbour ties
    elif case == "special":
        pick = torch.rand(B * N, C, generator=g)
        r = torch.where(pick < 0.02, torch.tensor(float("nan")), r)
        r = torch.where((pick >= 0.02) & (pick < 0.04), torch.tensor(float("inf")), r)
        r = torch.where((pick >= 0.04) & (pick < 0.06), torch.tensor(float("-inf")), r)
        r = torch.where((pick >= 0.06) & (pick < 0.10), torch.tensor(-0.0), r)
    r = r.to(BF).to(DEV)
    idx = torch.randint(0, N, (B, N, k), generator=g).to(torch.int32).to(DEV)
    aff = None if case == "noaffine" else ops.BNAffine(sc.to(DEV), sh.to(DEV))
    launch_counters(reset=True)
    u_key, a_key = ops.mr_aggregate_fwd(r, idx, B, N, C, aff)
    assert launch_counters()["mr_fwd_key"] == 1
    try:
        ops.set_tuning("mr_key_min_k", 0)
        launch_counters(reset=True)
        u_ref, a_ref = ops.mr_aggregate_fwd(r, idx, B, N, C, aff)
        assert launch_counters()["mr_fwd_key"] == 0 and launch_counters()["mr_fwd_lds"] == 1
    finally:
        ops.reset_tuning()
    assert torch.equal(a_key, a_ref), float((a_key != a_ref).float().mean())
    assert torch.equal(u_key.view(torch.int16), u_ref.view(torch.int16))          # bit patterns: NaN == NaN, -0 != +0


@pytest.mark.parametrize("N,C,k,dt", [(256, 64, 18, "bf16"), (128, 128, 18, "bf16"), (64, 256, 3, "bf16"), (32, 512, 5, "bf16"),
                                       (64, 256, 18, "bf16"), (32, 512, 18, "bf16"), (128, 128, 18, "fp32"), (256, 64, 3, "fp32")])
def test_max_relative_backward_with_hub_nodes(ops, N, C, k, dt):
    """csrc/mr.hip mr_bwd_kernel (CSR gather over the reversed graph) against a dense fp64 evaluation of torch_vertex.py:21-32's backward
    on graphs with hub nodes (a third of the ids point at node 0: in-degrees of hundreds) and repeated ids inside a list, arg-max
    bytes from the forward kernel"""
    B = 32
    tdt = BF if dt == "bf16" else torch.float32
    g = torch.Generator().manual_seed(N + C + k)
    r = torch.randn(B * N, C, generator=g).to(tdt).to(DEV)
    idx = torch.randint(0, N, (B, N, k), generator=g)
    idx = torch.where(torch.rand(B, N, k, generator=g) < 0.33, torch.zeros_like(idx), idx).to(torch.int32).to(DEV)
    du = torch.randn(B * N, 2 * C, generator=g).to(tdt).to(DEV)
    _, amax = ops.mr_aggregate_fwd(r, idx, B, N, C)
    dy = ops.mr_aggregate_bwd(du, idx, amax, B, N, C)
    d = du.double().reshape(B, N, C, 2)
    ref = d[..., 0] - d[..., 1]
    tgt = torch.gather(idx.long(), 2, amax.reshape(B, N, C).long())                       # (B, N, C): the row each element's gradient goes to
    ref = ref.scatter_add(1, tgt, d[..., 1].contiguous()).reshape(B * N, C)
    assert relerr(dy, ref) < (4e-3 if dt == "bf16" else 1e-6), relerr(dy, ref)      # the output is rounded to the storage type once


@pytest.mark.parametrize("N,C", [(256, 64), (128, 128), (64, 256), (32, 512)])
@pytest.mark.parametrize("k", [9, 18])
@pytest.mark.parametrize("graph", ["uniform", "hub", "one_target", "out_of_range"])
def test_degree_ranked_backward_equals_the_plain_gather(ops, N, C, k, graph):
    """csrc/mr.hip mr_bwd_sorted_kernel (deep plan: nodes ranked by in-degree, SDWA select) against mr_bwd_kernel on the same inputs:
    the same fp32 sums over the reversed edge list (its order comes from LDS atomics in both, so a sum may differ in its last fp32 bits
    before the one rounding: at most one bf16 ulp apart), and against a dense fp64 evaluation of torch_vertex.py:21-32's backward"""
    B = 24
    g = torch.Generator().manual_seed(7 * N + C + k)
    r = torch.randn(B * N, C, generator=g).to(BF).to(DEV)
    idx = torch.randint(0, N, (B, N, k), generator=g)
    if graph == "hub":
        idx = torch.where(torch.rand(B, N, k, generator=g) < 0.4, torch.full_like(idx, 3), idx)
    elif graph == "one_target":
        idx = torch.full_like(idx, N - 1)                          # in-degree N * k at one node, 0 elsewhere
    elif graph == "out_of_range":
        idx = idx + torch.randint(-2, 3, idx.shape, generator=g) * N    # the kernels clamp ids, as the forward does
    idx = idx.to(torch.int32).to(DEV)
    du = torch.randn(B * N, 2 * C, generator=g).to(BF).to(DEV)
    _, amax = ops.mr_aggregate_fwd(r, idx, B, N, C)
    try:
        ops.set_tuning("mr_bwd_sorted_min_k", 8)
        ops.launch_counters(reset=True)
        got = ops.mr_aggregate_bwd(du, idx, amax, B, N, C)
        torch.cuda.synchronize()
        assert ops.launch_counters()["mr_bwd_sorted"] == 1
        ops.set_tuning("mr_bwd_sorted_min_k", 0)
        ops.launch_counters(reset=True)
        plain = ops.mr_aggregate_bwd(du, idx, amax, B, N, C)
        torch.cuda.synchronize()
        assert ops.launch_counters()["mr_bwd_sorted"] == 0
    finally:
        ops.reset_tuning()
    a, b = got.float(), plain.float()
    ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-30
    assert bool(((a - b).abs() <= ulp).all())
    assert (a != b).float().mean().item() < 0.02
    d = du.double().reshape(B, N, C, 2)
    tgt = torch.gather(idx.long().clamp(0, N - 1), 2, amax.reshape(B, N, C).long())
    ref = (d[..., 0] - d[..., 1]).scatter_add(1, tgt, d[..., 1].contiguous()).reshape(B * N, C)
    assert relerr(got, ref) < 4e-3, relerr(got, ref)


@pytest.mark.parametrize("N,C", [(256, 64), (128, 128), (64, 256), (32, 512)])
@pytest.mark.parametrize("k,act", [(3, 0), (9, 1), (18, 2)])
def test_aggregation_backward_emits_the_batchnorm_column_sums(ops, N, C, k, act):
    """nsid_mr_aggregate_bwd_bn: the degree-ranked aggregation backward also returns step 1 of the backward of the BatchNorm in front
    of the aggregation (Grapher fc1, torch_vertex.py:183-195), one row of sums per clip, taken from its own ROUNDED dy: the same dy bits
    as the plain call, and sums equal to nsid_bn_bwd_reduce on that dy (fp64 evaluation: fp32 summation noise)"""
    B = 16
    g = torch.Generator().manual_seed(11 * N + C + k)
    r = torch.randn(B * N, C, generator=g).to(BF).to(DEV)
    idx = torch.randint(0, N, (B, N, k), generator=g)
    idx = torch.where(torch.rand(B, N, k, generator=g) < 0.3, torch.full_like(idx, 1), idx).to(torch.int32).to(DEV)
    du = torch.randn(B * N, 2 * C, generator=g).to(BF).to(DEV)
    scale = 1 + 0.2 * torch.randn(C, generator=g)
    scale[::3] = -scale[::3]
    shift, mean, invstd = 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    aff = ops.BNAffine(scale.to(DEV), shift.to(DEV), mean.to(DEV), invstd.to(DEV))
    _, amax = ops.mr_aggregate_fwd(r, idx, B, N, C)
    ops.launch_counters(reset=True)
    dy, part = ops.mr_aggregate_bwd(du, idx, amax, B, N, C, bn=(r, aff, act))
    torch.cuda.synchronize()
    assert part is not None and tuple(part.shape) == (2, B, C) and ops.launch_counters()["mr_bwd_sorted"] == 1
    plain = ops.mr_aggregate_bwd(du, idx, amax, B, N, C)
    a, b_ = dy.float(), plain.float()
    assert bool(((a - b_).abs() <= torch.maximum(a.abs(), b_.abs()) * 2.0 ** -7 + 1e-30).all())
    d, x = dy.double().cpu().reshape(B, N, C), r.double().cpu().reshape(B, N, C)
    slope = {0: 1.0, 1: 0.0, 2: 0.2}[act]
    z = scale.double() * x + shift.double()
    gg = torch.where(z > 0, d, d * slope) if act else d
    xhat = (x - mean.double()) * invstd.double()
    s0, s1 = gg.sum(1), (gg * xhat).sum(1)
    a0, a1 = gg.abs().sum(1), (gg * xhat).abs().sum(1)
    pc = part.double().cpu()
    assert ((pc[0] - s0).abs() <= 4e-6 * a0 + 1e-5).all(), float((pc[0] - s0).abs().max())
    assert ((pc[1] - s1).abs() <= 4e-6 * a1 + 1e-5).all(), float((pc[1] - s1).abs().max())
    # and the whole layer: BatchNorm backward with these sums == with the reduce pass's
    dgam = [torch.zeros(C, device=DEV) for _ in range(2)]
    dbet = [torch.zeros(C, device=DEV) for _ in range(2)]
    dr_a = ops.bn_backward(dy, r, aff, act, dgam[0], dbet[0], partial=part)
    dr_b = ops.bn_backward(dy, r, aff, act, dgam[1], dbet[1])
    assert relerr(dgam[0], dgam[1]) < 1e-5 and relerr(dbet[0], dbet[1]) < 1e-5
    assert relerr(dr_a, dr_b) < 2e-3


def test_eval_ffn256_variants_agree(ops):
    """csrc/ffn256_fused.hip under its tuning key: persistent over the row tiles (300 tiles on one workgroup per CU: some take two, the
    next tile's x fetched by the epilogue of the one before), one workgroup per tile, and the 4-wave form with the output accumulators
    in AGPRs all evaluate the same sums in the same order -- bit-identical outputs"""
    M, C, H = 76800, 256, 1024
    x = synth_randn(f"ffx{M}{C}", M, C).to(BF).to(DEV)
    w1 = (synth_randn(f"ffw1{C}", H, C) * C ** -0.5).to(DEV)
    w2 = (synth_randn(f"ffw2{C}", C, H) * H ** -0.5).to(DEV)
    b1, b2 = (0.3 * synth_randn(f"ffb1{C}", H)).to(DEV), (0.3 * synth_randn(f"ffb2{C}", C)).to(DEV)
    for w in (w1, w2):
        ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    outs = {}
    try:
        for key in (1, 2, 4):
            ops.set_tuning("ffn256", key)
            outs[key] = ops.ffn_fused_fwd(x, w1, b1, w2, b2, M, C, H)
            torch.cuda.synchronize()
    finally:
        ops.set_tuning("ffn256", 1)
    assert all(o is not None for o in outs.values())
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[4])


@pytest.mark.parametrize("M,Nout,K,groups,affine", [(512, 256, 512, 1, True), (200, 64, 64, 1, False), (384, 128, 128, 4, True)])
def test_linear_fwd_with_residual_addend(ops, M, Nout, K, groups, affine):
    """nsid_linear_fwd_res: out = f(x) W^T + bias + addend in one launch (eval path: conv + folded BatchNorm + shortcut)"""
    x = synth_randn(f"rx{M}{K}", M, groups * K).to(BF)
    w = synth_randn(f"rw{Nout}{K}", groups * Nout, K) * K ** -0.5
    bias = synth_randn(f"rb{Nout}", groups * Nout)
    add = synth_randn(f"ra{M}{Nout}", M, groups * Nout).to(BF)
    sc = 1 + 0.2 * synth_randn("rsc", groups * K) if affine else None
    sh = 0.3 * synth_randn("rsh", groups * K) if affine else None
    xin = act_ref(x.float() * sc + sh, 1) if affine else x.float()
    d = lambda t: None if t is None else t.to(DEV)
    ref = torch.cat([bfr(xin[:, g * K:(g + 1) * K]) @ bfr(w[g * Nout:(g + 1) * Nout]).t() for g in range(groups)], 1)
    ref = ref + bias.double() + add.double()
    out, stat = ops.linear_fwd(d(x), d(w), d(bias), M, Nout, K, groups, d(sc), d(sh), 1 if affine else 0, 0, addend=d(add))
    assert stat is None and out.dtype == BF and relerr(out, ref) < 2.5e-3
    with pytest.raises(ValueError):
        ops.linear_fwd(d(x), d(w), d(bias), M, Nout, K, groups, addend=d(add), want_stat=True)


def test_eval_batchnorm_folding_matches_the_unfolded_path(ops, golden):
    """fingerprint extraction with every eval-mode BatchNorm folded into its conv (and the shortcut in the GEMM epilogue)
    = the conv / BatchNorm / shortcut passes kept apart, up to bf16 rounding of the folded weights"""
    from neuralsampleid_amd import functional as F_
    F_.set_activation_dtype("bf16")
    g = golden("e2e_b8_k3")
    model = build()
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).eval()
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    n = len([k for k in g if k.startswith("knn.eval.")])
    gold_idx = [g.t(f"knn.eval.{c}") for c in range(n)]       # same neighbour sets on both sides (kNN near-ties)
    outs = {}
    ops._FOLDED.clear()
    try:
        for fold in (False, True):
            F_.FOLD_EVAL_BN = fold
            F_.TAPE = F_.KnnTape(replay=gold_idx)
            with torch.no_grad():
                outs[fold] = model(x_i, x_j)
            assert (len(ops._FOLDED) > 50) == fold         # the folded path really ran (64 BatchNorm layers), only then
    finally:
        F_.TAPE = None
        F_.FOLD_EVAL_BN = True
    for a, b in zip(outs[True], outs[False]):
        assert relerr(a, b) < 2e-2, relerr(a, b)
    cos = torch.nn.functional.cosine_similarity(outs[True][2], outs[False][2], dim=1)
    assert float(cos.min()) > 0.9995, float(cos.min())
    # and the folded path meets the reference goldens like the unfolded one does
    assert relerr(outs[True][0], g.t("h_i_eval")) < 4e-2


def test_lds_dma_gemm_inside_the_eval_forward(ops, golden):
    """the folded eval forward with EVERY eligible GEMM on csrc/gemm256.hip (threshold forced to one tile: the C >= 128 stages' fc1 / fc2 /
    FFN layers take the plain, ReLU and residual epilogues) against the same forward on csrc/gemm.hip and against the reference goldens"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd._lib import call, lib, reset_tuning, set_tuning
    F_.set_activation_dtype("bf16")
    g = golden("e2e_b8_k3")
    model = build()
    model.load_state_dict(synth_state(model.state_dict()))
    model.to(DEV).eval()
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    n = len([k for k in g if k.startswith("knn.eval.")])
    gold_idx = [g.t(f"knn.eval.{c}") for c in range(n)]
    outs, launches = {}, {}
    try:
        for thr in (0, 1):
            set_tuning("g256_min", thr)
            n0 = lib.nsid_gemm_g256_launches()
            F_.TAPE = F_.KnnTape(replay=gold_idx)
            with torch.no_grad():
                outs[thr] = model(x_i, x_j)
            torch.cuda.synchronize()
            launches[thr] = lib.nsid_gemm_g256_launches() - n0
    finally:
        F_.TAPE = None
        reset_tuning()
    assert launches[0] == 0 and launches[1] >= 2 * 16, launches          # two views x (10 blocks with C >= 128: FFN fc1 + fc2 at least)
    for a, b in zip(outs[1], outs[0]):
        assert relerr(a, b) < 1e-2, relerr(a, b)                          # same operands and rounding points, another summation order
    cos = torch.nn.functional.cosine_similarity(outs[1][2], outs[0][2], dim=1)
    assert float(cos.min()) > 0.9998, float(cos.min())
    assert relerr(outs[1][0], g.t("h_i_eval")) < 4e-2


@pytest.mark.parametrize("M,Nout,K,relu,stat,res,bias", [(16384, 1024, 256, False, True, False, False),
                                                          (16384, 1024, 256, False, True, False, True),
                                                          (8192, 2048, 512, False, True, False, False),
                                                          (16384, 256, 1024, False, False, True, True),
                                                          (16384, 256, 1024, True, False, False, True),
                                                          (1024, 256, 128, False, False, True, True), (1024, 256, 128, False, True, False, True),
                                                          (512, 512, 2048, True, False, False, False)])
def test_lds_dma_256_tile_forward(ops, M, Nout, K, relu, stat, res, bias):
    """csrc/gemm256.hip (tuning keys g256_min / g256_train): 256x256 tiles staged by LDS-DMA through a four-slot ring — product, bias,
    ReLU in the epilogue, residual addend and the BatchNorm statistics epilogue against an fp64 product of the bf16 operands.
    K = 128 is the shortest ring (prologue + tail only), K = 2048 runs 15 steady-state rounds."""
    from neuralsampleid_amd._lib import call, lib, reset_tuning, set_tuning
    g = torch.Generator().manual_seed(31)
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(Nout, K, generator=g) * K ** -0.5).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    b = torch.randn(Nout, generator=g).to(DEV) if bias else None
    add = torch.randn(M, Nout, generator=g).to(BF).to(DEV) if res else None
    ref = x.double() @ w.to(BF).double().t() + (b.double() if bias else 0)
    if relu:
        ref = ref.clamp_min(0)
    n0 = lib.nsid_gemm_g256_launches()
    set_tuning("g256_min", 1); set_tuning("g256_train", 1)
    try:
        out, st = ops.linear_fwd(x, w, b, M, Nout, K, 1, None, None, 0, 1 if relu else 0, want_stat=stat, addend=add)
        torch.cuda.synchronize()
    finally:
        reset_tuning()
    assert lib.nsid_gemm_g256_launches() == n0 + 1                       # the kernel under test really ran
    # the only rounding is the bf16 store of the fp32 result
    full = ref + (add.double() if res else 0)
    err = (out.double() - full).abs()
    assert float((err / (full.abs() + 1.0)).max()) < 2 ** -8, float((err / (full.abs() + 1.0)).max())
    assert relerr(out, full) < 2.5e-3
    if stat:
        tiles = ref.reshape(M // 128, 128, Nout)
        assert relerr(st[0], tiles.sum(1)) < 1e-4 and relerr(st[1], (tiles * tiles).sum(1)) < 1e-4


@pytest.mark.parametrize("M,Nout,K,groups", [(16384, 1024, 256, 1), (8192, 128, 128, 4), (640, 72, 40, 1), (131072, 256, 64, 1)])
def test_forward_relu_epilogue(ops, M, Nout, K, groups):
    """act_out = ReLU in the store path of every forward tile shape (128x128, grouped 128x64, ragged, gemm256.hip's 256x256): the values an
    eval-mode consumer used to obtain by ReLU on load"""
    g = torch.Generator().manual_seed(41)
    x = torch.randn(M, groups * K, generator=g).to(BF).to(DEV)
    w = (torch.randn(groups * Nout, K, generator=g) * K ** -0.5).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    b = torch.randn(groups * Nout, generator=g).to(DEV)
    out, _ = ops.linear_fwd(x, w, b, M, Nout, K, groups, None, None, 0, 1)
    plain, _ = ops.linear_fwd(x, w, b, M, Nout, K, groups)
    assert torch.equal(out, plain.clamp_min(0))              # ReLU commutes with the bf16 rounding of the store
    assert float(out.float().min()) == 0.0


@pytest.mark.parametrize("M,Nout,K,affine,res", [(16384, 1024, 256, False, False), (8192, 2048, 512, True, True)])
def test_large_forward_without_statistics_on_gemm_hip(ops, M, Nout, K, affine, res):
    """eval-mode forward GEMMs with >= 1024 tiles of 128x128 and bf16 weight shadows kept on csrc/gemm.hip (g256_min = 0; its 256x128
    forms were removed in round 4: every instantiation spilled): same results as the reference product, with and without the residual
    addend, one workgroup per 128x128 tile"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(Nout, K, generator=g) * K ** -0.5).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    bias = torch.randn(Nout, generator=g).to(DEV)
    sc = (1 + 0.2 * torch.randn(K, generator=g)).to(DEV) if affine else None
    sh = (0.3 * torch.randn(K, generator=g)).to(DEV) if affine else None
    add = torch.randn(M, Nout, generator=g).to(BF).to(DEV) if res else None
    xin = act_ref(x.float() * sc + sh, 1) if affine else x.float()
    ref = xin.to(BF).double() @ w.to(BF).double().t() + bias.double()
    if res:
        ref = ref + add.double()
    from neuralsampleid_amd._lib import call, lib, reset_tuning, set_tuning
    trace = torch.zeros(4 * 8192, dtype=torch.int64, device=DEV)          # one record per workgroup (nsid_debug_gemm_trace)
    set_tuning("g256_min", 0)                 # (this test is about gemm.hip)
    assert lib.nsid_debug_gemm_trace(trace.data_ptr()) == 0
    try:
        out, stat = ops.linear_fwd(x, w, bias, M, Nout, K, 1, sc, sh, 1 if affine else 0, 0, want_stat=False, addend=add)
        torch.cuda.synchronize()
    finally:
        lib.nsid_debug_gemm_trace(None)
        reset_tuning()
    assert int((trace.view(-1, 4)[:, 0] != 0).sum()) == (M // 128) * (Nout // 128)      # 128x128 tiles
    assert stat is None and relerr(out, ref) < 2.5e-3
    # the statistics epilogue keeps the 128-row tiles (training path): per-tile sums still match
    out2, stat2 = ops.linear_fwd(x, w, bias, M, Nout, K, 1, sc, sh, 1 if affine else 0, 0, want_stat=True)
    tiles = (ref - (add.double() if res else 0)).reshape(M // 128, 128, Nout)
    assert relerr(stat2[0], tiles.sum(1)) < 1e-4 and relerr(out2, tiles.reshape(M, Nout)) < 2.5e-3


@pytest.mark.parametrize("M,Nout,K,res", [(512, 256, 512, True), (16384, 1024, 256, False), (256, 128, 64, False)])
def test_linear_fwd_relu_on_load_without_affine(ops, M, Nout, K, res):
    """activation-on-load with no affine (eval path: the producer's BatchNorm is folded into its weights): ReLU applied on the
    packed bf16 operand values == relu(x) W^T"""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    x[0, :8] = torch.tensor([-0.0, 0.0, -1.5, 2.0, -3e-39, 1e-38, -65504.0, 7.0]).to(BF)      # signed zeros, tiny values
    w = (torch.randn(Nout, K, generator=g) * K ** -0.5).to(DEV)
    ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    bias = torch.randn(Nout, generator=g).to(DEV)
    add = torch.randn(M, Nout, generator=g).to(BF).to(DEV) if res else None
    ref = torch.relu(x.double()) @ w.to(BF).double().t() + bias.double()
    if res:
        ref = ref + add.double()
    out, _ = ops.linear_fwd(x, w, bias, M, Nout, K, 1, None, None, 1, 0, addend=add)
    assert relerr(out, ref) < 2.5e-3
    with pytest.raises(ValueError):                       # LeakyReLU / ragged shapes have no such kernel: loud, not silent
        ops.linear_fwd(x, w, bias, M, Nout, K, 1, None, None, 2, 0)
