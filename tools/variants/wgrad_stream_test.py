"""Streaming weight gradients (csrc/wgrad_stream.hip) against an fp64 evaluation of the same bf16 operands and against the staged
forms of csrc/gemm.hip / csrc/wgrad.hip they replace (reference: the backward of every Conv2d 1x1 at
encoder/gcn_lib/torch_vertex.py:152-162, encoder/graph_encoder.py:74-77 whose weight matrix is a multiple of 128 x 128).

The arithmetic is the staged forms': bf16 operands (a producer BatchNorm + activation evaluated in fp32 and rounded once), fp32
accumulation, fp32 atomics over the row splits -- so the result is held to the fp64 value within fp32 summation noise, and the two
implementations to each other within twice that."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (M, Nout, K, groups): the weight gradients of the C = 128 / 256 / 512 stages (rows reduced), plus row counts that exercise the
# 1-, 2- and 3-stage tails of the ring and a split that does not divide evenly into the workgroup target
SHAPES = [(2048, 256, 1024, 1), (2048, 1024, 256, 1), (4096, 128, 128, 4), (1024, 512, 512, 1), (4096, 128, 256, 1),
          (64, 128, 128, 1), (128, 128, 128, 1), (192, 256, 128, 1), (1600, 128, 128, 1), (8192, 256, 256, 4)]


@pytest.fixture()
def bf16_mode():
    from neuralsampleid_amd import ops
    prec = ops.get_gemm_precision()
    ops.set_gemm_precision("bf16")
    ops.reset_tuning()
    yield ops
    ops.reset_tuning()
    ops.set_gemm_precision(prec)


def _operands(M, N, K, G, aff, seed=0):
    g = torch.Generator(device="cpu").manual_seed(4000 + seed)
    dy = torch.randn(M, G * N, generator=g).to(torch.bfloat16)
    x = (torch.randn(M, G * K, generator=g) * 1.5).to(torch.bfloat16)
    sc = (1 + 0.2 * torch.randn(G * K, generator=g)) if aff else None
    sh = (0.3 * torch.randn(G * K, generator=g)) if aff else None
    if aff:
        sc[::5] = -sc[::5]
    return dy, x, sc, sh


def _fp64(dy, x, sc, sh, slope, M, N, K, G):
    xf = x.float()
    if sc is not None:
        v = (sc.double() * xf.double() + sh.double()).float()
        v = torch.maximum(v, v * slope)
        xf = v.to(torch.bfloat16).float()
    out = torch.empty(G * N, K, dtype=torch.float64)
    for g in range(G):
        out[g * N:(g + 1) * N] = dy[:, g * N:(g + 1) * N].double().t() @ xf[:, g * K:(g + 1) * K].double()
    return out


def _run(ops, dy, x, sc, sh, act, M, N, K, G, dw0):
    dw = dw0.clone().to(DEV)
    d = lambda t: None if t is None else t.to(DEV)
    ops.linear_bwd_weight(d(dy), d(x), dw, M, N, K, G, d(sc), d(sh), act)
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("M,N,K,G", SHAPES)
@pytest.mark.parametrize("act", [None, "relu", "leaky"])
@pytest.mark.parametrize("slots", [3, 4])
def test_streaming_weight_gradient_matches_fp64_and_the_staged_forms(bf16_mode, M, N, K, G, act, slots):
    ops = bf16_mode
    aff = act is not None
    code, slope = {None: (ops.ACT_NONE, 1.0), "relu": (ops.ACT_RELU, 0.0), "leaky": (ops.ACT_LEAKY, 0.2)}[act]
    dy, x, sc, sh = _operands(M, N, K, G, aff, seed=M + N + K + G)
    ref = _fp64(dy, x, sc, sh, slope, M, N, K, G)
    dw0 = torch.randn(G * N, K)                                   # the kernel ACCUMULATES (the second view adds to the first)
    ops.set_tuning("wgrad_stream", 1)
    ops.set_tuning("wgs_min_wgs", 1)
    ops.set_tuning("wgs_min_rows", 64)
    ops.set_tuning("wgs_slots", slots)
    ops.launch_counters(reset=True)
    got = _run(ops, dy, x, sc, sh, code, M, N, K, G, dw0)
    assert ops.launch_counters()["wgrad_stream"] == 1, "the streaming form did not take this shape"
    ops.set_tuning("wgrad_stream", 0)
    ops.launch_counters(reset=True)
    staged = _run(ops, dy, x, sc, sh, code, M, N, K, G, dw0)
    assert ops.launch_counters()["wgrad_stream"] == 0
    scale = ref.abs().max().item()
    # fp32 accumulation of M products of magnitude ~1.5: noise ~ sqrt(M) * 2^-24 * |terms|; both forms sit far inside 1e-5 * scale * ...
    tol = 2e-6 * scale + 1e-5
    assert (got.double() - dw0.double() - ref).abs().max().item() <= tol * 4
    assert (staged.double() - dw0.double() - ref).abs().max().item() <= tol * 4
    assert (got - staged).abs().max().item() <= tol * 8


def test_streaming_weight_gradient_row_split_choice(bf16_mode):
    """the launcher's split choice: at least wgs_min_rows rows per workgroup, whole 64-row stages, never more workgroups than the target"""
    ops = bf16_mode
    M, N, K, G = 6 * 64, 128, 128, 1                              # 6 stages: splits of 1, 2, 3, 6 stages are possible
    dy, x, _, _ = _operands(M, N, K, G, False)
    ref = _fp64(dy, x, None, None, 1.0, M, N, K, G)
    ops.set_tuning("wgs_min_wgs", 1)
    for target, min_rows in [(1, 64), (4, 64), (6, 64), (64, 64), (64, 128), (64, 384)]:
        ops.set_tuning("wgs_wgs", target)
        ops.set_tuning("wgs_min_rows", min_rows)
        got = _run(ops, dy, x, None, None, ops.ACT_NONE, M, N, K, G, torch.zeros(N, K))
        assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-5, (target, min_rows)


def test_streaming_weight_gradient_refuses_other_shapes(bf16_mode):
    ops = bf16_mode
    for M, N, K, G in [(1024, 64, 128, 1), (1024, 128, 64, 1), (1000, 128, 128, 1), (1024, 192, 128, 1)]:
        dy, x, _, _ = _operands(M, N, K, G, False)
        ref = _fp64(dy, x, None, None, 1.0, M, N, K, G)
        ops.launch_counters(reset=True)
        got = _run(ops, dy, x, None, None, ops.ACT_NONE, M, N, K, G, torch.zeros(G * N, K))
        assert ops.launch_counters()["wgrad_stream"] == 0
        assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-5
