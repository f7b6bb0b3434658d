"""Run-to-run reproducibility: the same kNN launch repeated on the same operands must give the same bits, and the forward passes
(whose only atomics are the head's fp32 accumulations) the same values to an ulp or two. This is the hazard detector of the suite:
a missing wait state, an LDS ordering assumption or an instruction that misbehaves under load shows up as a launch that differs
from the first one, typically 0.1-1 % of the time on a handful of rows.
Motivation (round 3): the register-resident `knn_sel_kernel` passed every golden and still returned a wrong neighbour row in about
1 launch of 150 — hipcc had contracted `|y_i|^2 - 2 x` into `v_pk_fma_f32 ..., 2.0, v[si:si+1] op_sel:[0,0,1]`, and that instruction
intermittently dropped the |y_i|^2 term in lanes 48-63 (docs/experiments.md, tools/knn_sel_repro.py)."""
import os

import pytest
import torch

import tapes
from synth import GRAFP_CFG
from test_e2e_gpu import build_model, load_synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
REPEATS = int(os.environ.get("NSID_REPEATS", "150"))     # a soak run raises it (tools/gpu_fulltests.sh does not)


@pytest.fixture(autouse=True)
def _fp32_after():
    yield
    from neuralsampleid_amd import functional as F_
    F_.set_activation_dtype("fp32")
    F_.TAPE = None


def _capture_knn_calls(run):
    """every ops.knn_graph call of `run()` with its operands cloned"""
    from neuralsampleid_amd import ops
    calls, orig = [], ops.knn_graph

    def capture(r, B, N, C, k, dilation=1, aff=None):
        calls.append((r.clone(), B, N, C, k, dilation,
                      None if aff is None else ops.BNAffine(aff.scale.clone(), aff.shift.clone())))
        return orig(r, B, N, C, k, dilation, aff)
    ops.knn_graph = capture
    try:
        run()
    finally:
        ops.knn_graph = orig
    return calls


def _repeat_knn(calls):
    from neuralsampleid_amd import ops
    bad = []
    for ci, (r, B, N, C, k, d, aff) in enumerate(calls):
        first = ops.knn_graph(r, B, N, C, k, d, aff)
        differ = torch.zeros((), dtype=torch.int32, device=DEV)
        for _ in range(REPEATS):
            differ += (ops.knn_graph(r, B, N, C, k, d, aff) != first).any().int()
        if int(differ):
            bad.append((ci, N, C, k, d, int(differ)))
    return bad


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_deep_configuration_knn_kernels_repeat_bitwise(golden, mode):
    """knn_sel_kernel (N = 256, k d = 18) and knn_rank_kernel (k d = 36 / 54 / 18) on the deep configuration's own train-mode
    features (forward of tests/golden/deep_b4_k18, the reference's neighbour ids teacher-forced so that the fp32 features are the
    golden's; bf16 storage = the instantiations bench.py --deep times): 48 calls x 150 launches"""
    from neuralsampleid_amd import functional as F_
    F_.set_activation_dtype(mode)
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden("deep_b4_k18")
    model = load_synth(SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=18, size="t",
                                                      blocks=[4, 4, 12, 4], use_dilation=True))).train()
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    # the reference's step-0 graphs: one strict-fp32 forward, own search + the fixture's near-tie rows (tests/tapes.py)
    F_.set_activation_dtype("fp32")
    F_.TAPE = tape = F_.KnnTape(patch=tapes.patches_of(g, "s0"))
    with torch.no_grad():
        model(x_i, x_j)
    F_.TAPE = None
    assert tapes.check_patched(tape, g, "s0")[0] == 0
    gold_idx = [t.clone() for t in tape.patched]
    F_.set_activation_dtype(mode)

    def run():
        F_.TAPE = F_.KnnTape(replay=gold_idx)
        with torch.no_grad():
            model(x_i, x_j)
        F_.TAPE = None
    calls = _capture_knn_calls(run)
    assert len(calls) == 48 and {c[2] for c in calls} == {256, 128, 64, 32}
    assert _repeat_knn(calls) == []


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_knn2_kernel_repeats_bitwise(golden, mode):
    """the k = 3 kernels of the timed configuration, on the B = 8 golden's train-mode features, both storage types"""
    from neuralsampleid_amd import functional as F_
    F_.set_activation_dtype(mode)
    g = golden("e2e_b8_k3")
    model = build_model(3).train()
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)

    def run():
        with torch.no_grad():
            model(x_i, x_j)
    calls = _capture_knn_calls(run)
    assert len(calls) == 24
    assert _repeat_knn(calls) == []


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_forward_only_extraction_repeats_to_an_ulp(mode):
    """eval-mode forward of 512 clips (GEMMs incl. the 256x256-tile and fused FFN / aggregation forms, kNN, aggregation, head).
    Only the head accumulates with fp32 atomics (node mean / split reductions: +-1 ulp run to run, measured 6e-8 on unit-norm
    fingerprints); everything upstream — in particular what the kNN graphs are built from — is order-fixed, so 40 passes must
    agree to 1e-6: a dropped term or a stale tile moves a fingerprint by 1e-3 or more"""
    from neuralsampleid_amd import fingerprint, functional as F_
    F_.set_activation_dtype(mode)
    model = build_model(3).eval()
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(512, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gen) * 20 - 40).to(DEV)
    first = fingerprint.extract_fingerprints(model, x, 512).clone()
    worst = torch.zeros((), device=DEV)
    for _ in range(40):
        worst = torch.maximum(worst, (fingerprint.extract_fingerprints(model, x, 512) - first).abs().max())
    assert float(worst) <= 1e-6


def test_train_mode_forward_repeats_to_an_ulp():
    """train-mode forward (batch statistics from per-tile partial sums in a fixed order) at B = 64, bf16 storage: z of 30 passes
    within 2e-6 of the first (the head's atomics again)"""
    from neuralsampleid_amd import functional as F_
    F_.set_activation_dtype("bf16")
    model = build_model(3).train()
    gen = torch.Generator().manual_seed(6)
    x_i = (torch.randn(64, GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gen) * 20 - 40).to(DEV)
    x_j = x_i + 3 * torch.randn(x_i.shape, generator=gen).to(DEV)
    with torch.no_grad():
        first = torch.cat(model(x_i, x_j)[2:]).clone()
        worst = torch.zeros((), device=DEV)
        for _ in range(30):
            worst = torch.maximum(worst, (torch.cat(model(x_i, x_j)[2:]) - first).abs().max())
    assert float(worst) <= 2e-6


def _repeat_bitwise(fn, n=100):
    """fn() -> tuple of tensors; the number of repeats (of n) in which any output differs from the first call's"""
    first = [t.clone() for t in fn()]
    differ = torch.zeros((), dtype=torch.int32, device=DEV)
    for _ in range(n):
        for a, b in zip(fn(), first):
            differ += (a != b).any().int()
    return int(differ)


@pytest.mark.parametrize("M,Nout,K,groups", [(32768, 512, 128, 1), (32768, 128, 512, 1), (16384, 128, 128, 4), (8192, 2048, 512, 1)])
def test_gemm_family_repeats_bitwise_at_the_timed_shapes(M, Nout, K, groups):
    """forward (BatchNorm + ReLU on the operand load, statistics epilogue), backward-data (plain, and with the BatchNorm-backward
    column sums in the epilogue) and the BatchNorm-backward passes on bf16 activations at the step's own shapes: none of these
    accumulates with atomics, so 100 launches must agree bit for bit (the weight gradients do use atomics and are not here)"""
    from neuralsampleid_amd import functional as F_, ops
    F_.set_activation_dtype("bf16")
    g = torch.Generator(device=DEV).manual_seed(M + K)
    x = torch.randn(M, groups * K, device=DEV, generator=g).bfloat16()
    dy = torch.randn(M, groups * Nout, device=DEV, generator=g).bfloat16()
    w = torch.randn(groups * Nout, K, device=DEV, generator=g) * K ** -0.5
    sc = 1 + 0.1 * torch.randn(groups * K, device=DEV, generator=g)
    sh = 0.1 * torch.randn(groups * K, device=DEV, generator=g)
    aff = ops.BNAffine(sc, sh, mean=torch.zeros_like(sc), invstd=torch.ones_like(sc))
    assert _repeat_bitwise(lambda: ops.linear_fwd(x, w, None, M, Nout, K, groups, sc, sh, ops.ACT_RELU, 0, want_stat=True)) == 0
    assert _repeat_bitwise(lambda: (ops.linear_bwd_data(dy, w, M, Nout, K, groups),)) == 0
    assert _repeat_bitwise(lambda: ops.linear_bwd_data(dy, w, M, Nout, K, groups, bn=(x, aff, ops.ACT_RELU))) == 0
    dgamma, dbeta = torch.zeros(groups * K, device=DEV), torch.zeros(groups * K, device=DEV)
    dx = torch.randn(M, groups * K, device=DEV, generator=g).bfloat16()
    assert _repeat_bitwise(lambda: (ops.bn_backward(dx, x, aff, ops.ACT_RELU, dgamma, dbeta),)) == 0


@pytest.mark.parametrize("M,Nout,K,groups", [(32768, 512, 128, 1), (32768, 128, 128, 1), (16384, 128, 128, 4), (8192, 2048, 512, 1)])
def test_weight_gradients_repeat_to_atomics_noise(M, Nout, K, groups):
    """the weight-gradient kernels (rectangular / square gemm_kernel tiles, wgrad3_kernel) reduce their row splits with fp32 atomics:
    100 launches on the same operands agree to summation-order noise (1e-5 of the largest entry); a stale or dropped tile would be
    orders of magnitude above that"""
    from neuralsampleid_amd import functional as F_, ops
    F_.set_activation_dtype("bf16")
    g = torch.Generator(device=DEV).manual_seed(M + Nout)
    x = torch.randn(M, groups * K, device=DEV, generator=g).bfloat16()
    dy = torch.randn(M, groups * Nout, device=DEV, generator=g).bfloat16()
    sc = 1 + 0.1 * torch.randn(groups * K, device=DEV, generator=g)
    sh = 0.1 * torch.randn(groups * K, device=DEV, generator=g)

    def run():
        dw = torch.zeros(groups * Nout, K, device=DEV)
        ops.linear_bwd_weight(dy, x, dw, M, Nout, K, groups, sc, sh, ops.ACT_RELU)
        return dw
    first = run()
    worst = torch.zeros((), device=DEV)
    for _ in range(100):
        worst = torch.maximum(worst, (run() - first).abs().max())
    assert float(worst) <= 1e-5 * float(first.abs().max())


@pytest.mark.parametrize("N,C,k", [(256, 64, 3), (128, 128, 3), (64, 256, 18), (32, 512, 18)])
def test_aggregation_kernels_repeat(N, C, k):
    """max-relative aggregation at B = 256, bf16: the forward (values and arg-max bytes) bit for bit; the backward is a gather over a
    reversed graph whose per-node edge order depends on an integer-atomics race (mr.hip), i.e. fp32 sums of k terms on average in a
    varying order — results within one bf16 ulp of each other (order-dependent rounding of an fp32 sum), never a wrong edge"""
    from neuralsampleid_amd import functional as F_, ops
    F_.set_activation_dtype("bf16")
    B = 256
    g = torch.Generator(device=DEV).manual_seed(N + k)
    r = torch.randn(B * N, C, device=DEV, generator=g).bfloat16()
    idx = torch.randint(0, N, (B, N, k), device=DEV, generator=g, dtype=torch.int32)
    aff = ops.BNAffine(torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g) * 0.1)
    assert _repeat_bitwise(lambda: ops.mr_aggregate_fwd(r, idx, B, N, C, aff, True), n=50) == 0
    u, am = ops.mr_aggregate_fwd(r, idx, B, N, C, aff, True)
    du = torch.randn(B * N, 2 * C, device=DEV, generator=g).bfloat16()
    first = ops.mr_aggregate_bwd(du, idx, am, B, N, C).float()
    worst = torch.zeros((), device=DEV)
    for _ in range(50):
        worst = torch.maximum(worst, ((ops.mr_aggregate_bwd(du, idx, am, B, N, C).float() - first).abs() / first.abs().clamp_min(1.0)).max())
    assert float(worst) <= 2 ** -7          # one bf16 ulp of the result (order-dependent rounding of the fp32 sum), never a wrong edge


def test_ntxent_repeats_to_atomics_noise():
    """NT-Xent at the global batch of config 3 (2 048 pairs, this rank's 256): loss and dz to 1e-6 over 50 launches (split column ranges:
    partial (max, sum) merges and atomics into dz)"""
    from neuralsampleid_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    zi = torch.nn.functional.normalize(torch.randn(2048, 128, device=DEV, generator=g), dim=1)
    zj = torch.nn.functional.normalize(zi + 0.3 * torch.randn(2048, 128, device=DEV, generator=g), dim=1)
    l0, a0, b0 = [t.clone() for t in ops.ntxent_fwd_bwd(zi, zj, 0.05, 256, 256)]
    worst = torch.zeros((), device=DEV)
    for _ in range(50):
        l, a, b = ops.ntxent_fwd_bwd(zi, zj, 0.05, 256, 256)
        worst = torch.maximum(worst, torch.maximum((l - l0).abs().max(), torch.maximum((a - a0).abs().max(), (b - b0).abs().max())))
    assert float(worst) <= 1e-6 * max(1.0, float(a0.abs().max()))


@pytest.mark.parametrize("C,M", [(256, 131072), (128, 262144), (64, 524288)])
def test_register_tile_ffn_repeats_bitwise_under_load(C, M):
    """csrc/ffn256_fused.hip at the extraction micro-batch (two row tiles per persistent workgroup: the LDS-DMA ring, the per-chunk
    barrier / vmcnt hand-over and the next-tile x prefetch all run their second-tile paths), repeated REPEATS times while another
    stream keeps the chip busy with GEMMs (launch timing, L2 and LDS-DMA latency shift from launch to launch): the kernel has no
    atomics, so every launch must return the first one's bits — a slot overwritten early or read late would show as a differing tile"""
    from neuralsampleid_amd import ops
    ops.set_gemm_precision("bf16")
    H = 4 * C
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(H, C, generator=g) * C ** -0.5).to(DEV)
    w2 = (torch.randn(C, H, generator=g) * H ** -0.5).to(DEV)
    b1, b2 = (0.3 * torch.randn(H, generator=g)).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    for w in (w1, w2):
        ops.SHADOWS.register(w, ops.f32_to_bf16(w), owner=w)
    a = torch.randn(16384, 512, generator=g).to(torch.bfloat16).to(DEV)
    wa = (torch.randn(512, 512, generator=g) * 512 ** -0.5).to(DEV)
    ops.SHADOWS.register(wa, ops.f32_to_bf16(wa), owner=wa)
    first = ops.ffn_fused_fwd(x, w1, b1, w2, b2, M, C, H)
    assert first is not None
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = 0
    for r in range(REPEATS):
        with torch.cuda.stream(side):
            for _ in range(1 + r % 3):
                ops.linear_fwd(a, wa, None, 16384, 512, 512)
        out = ops.ffn_fused_fwd(x, w1, b1, w2, b2, M, C, H)
        bad += int(not torch.equal(out, first))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of {REPEATS} launches differ from the first"
