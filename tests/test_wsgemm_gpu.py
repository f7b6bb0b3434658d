"""Weight-stationary streaming GEMMs (csrc/wsgemm.hip) against an fp64 evaluation of the same bf16 operands, and against the tile
kernels of csrc/gemm.hip they replace (reference layers: encoder/gcn_lib/torch_vertex.py:152-162, encoder/graph_encoder.py:74-77,
encoder/gcn_lib/torch_nn.py:56 -- every Conv2d 1x1 whose whole weight matrix fits LDS).

The arithmetic is gemm.hip's -- bf16 operands, fp32 accumulation, one rounding of (acc + bias), statistics of the fp32 values -- with
another summation order (v_mfma_f32_32x32x16_bf16 instead of 16x16x32): outputs may differ from the tile kernels by one bf16 ulp where
the fp32 sums straddle a rounding boundary, so both are held to the fp64 value within that ulp, and the statistics to fp32 noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (M, Nout, K, groups, affine): the forward launches of one view at the timed batch (M reduced), tools/gemm_bench.py's table
FWD_SHAPES = [(512, 64, 64, 1, False), (384, 32, 32, 4, False), (256, 64, 128, 1, True), (640, 256, 64, 1, False),
              (256, 64, 256, 1, True), (384, 128, 128, 1, False), (256, 64, 64, 4, False), (384, 128, 256, 1, True),
              (256, 128, 128, 4, False), (128, 64, 64, 1, True)]


@pytest.fixture()
def ws_mode():
    from neuralsampleid_amd import ops
    prec = ops.get_gemm_precision()
    ops.set_gemm_precision("bf16")
    yield ops
    ops.set_tuning("ws_gemm", ops.get_tuning("ws_gemm"))
    ops.reset_tuning()
    ops.set_gemm_precision(prec)


def _operands(M, N, K, G, aff, seed=0):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    x = (torch.randn(M, G * K, generator=g) * 1.5).to(torch.bfloat16)
    w = torch.randn(G * N, K, generator=g) * K ** -0.5
    bias = 0.3 * torch.randn(G * N, generator=g)
    sc = (1 + 0.2 * torch.randn(G * K, generator=g)) if aff else None
    sh = (0.3 * torch.randn(G * K, generator=g)) if aff else None
    if aff:
        sc[::7] = -sc[::7]                      # negative BatchNorm scales occur (gamma < 0)
    return x, w, bias, sc, sh


def _fp64_forward(x, w, bias, sc, sh, act, M, N, K, G):
    """what the kernels compute, in fp64: operands as the MFMA sees them (bf16 x, affine + activation in fp32 then bf16, bf16 weights)"""
    xf = x.float()
    if sc is not None:
        v = (sc.double() * xf.double() + sh.double()).float()        # one fma per element, as the kernel
        v = torch.maximum(v, v * {1: 0.0, 2: 0.2}[act])
        v = torch.where(torch.isnan(xf), xf, v)
        xf = v.to(torch.bfloat16).float()
    wb = w.to(torch.bfloat16).double()
    out = torch.empty(M, G * N, dtype=torch.float64)
    for g in range(G):
        out[:, g * N:(g + 1) * N] = xf[:, g * K:(g + 1) * K].double() @ wb[g * N:(g + 1) * N].t()
    return out + (bias.double() if bias is not None else 0.0)


@pytest.mark.parametrize("M,N,K,G,aff", FWD_SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_ws_forward_matches_fp64_and_the_tile_kernel(ws_mode, M, N, K, G, aff, with_bias):
    ops = ws_mode
    x, w, bias, sc, sh = _operands(M, N, K, G, aff, seed=M + N + K + G)
    if not with_bias:
        bias = None
    act = ops.ACT_RELU if aff else ops.ACT_NONE
    ref = _fp64_forward(x, w, bias, sc, sh, 1, M, N, K, G)
    xd, wd = x.to(DEV), w.to(DEV)
    bd = bias.to(DEV) if bias is not None else None
    scd, shd = (sc.to(DEV), sh.to(DEV)) if aff else (None, None)
    res = {}
    for mode in (0, 1):
        ops.set_tuning("ws_gemm", mode)
        ops.launch_counters(reset=True)
        out, stat = ops.linear_fwd(xd, wd, bd, M, N, K, G, scd, shd, act, ops.ACT_NONE, want_stat=True)
        torch.cuda.synchronize()
        assert ops.launch_counters()["ws_fwd"] == mode, "the weight-stationary form must (not) have run"
        res[mode] = (out.float().cpu().double(), stat.cpu().double())
    # half a bf16 ulp (<= 2^-8 |v|) is the rounding itself; the absolute term covers fp32 accumulation noise that moves a sum across a
    # rounding boundary, and an operand whose fp32 affine differs from the fp64-then-fp32 evaluation here by an ulp
    ulp = ref.abs() * 2.0 ** -8 + 2e-3
    for mode in (0, 1):
        out, stat = res[mode]
        assert ((out - ref).abs() <= ulp).all(), (mode, float(((out - ref).abs() / ulp).max()))
        tiles = M // 128
        s_ref = ref.reshape(tiles, 128, G * N).sum(1)
        q_ref = (ref * ref).reshape(tiles, 128, G * N).sum(1)
        a_ref = ref.abs().reshape(tiles, 128, G * N).sum(1)
        assert stat.shape == (2, tiles, G * N)
        assert ((stat[0] - s_ref).abs() <= 3e-6 * a_ref + 1e-6).all()
        assert ((stat[1] - q_ref).abs() <= 3e-6 * q_ref + 1e-6).all()
    # the two kernels agree except where an fp32 sum sits on a bf16 rounding boundary
    diff = (res[0][0] != res[1][0]).double().mean()
    assert diff < 2e-2, float(diff)
    # and without statistics (eval-mode callers): the same outputs
    ops.set_tuning("ws_gemm", 1)
    out2, stat2 = ops.linear_fwd(xd, wd, bd, M, N, K, G, scd, shd, act, ops.ACT_NONE, want_stat=False)
    assert stat2 is None and torch.equal(out2.float().cpu().double(), res[1][0])


def test_ws_forward_leaky_activation_and_nan_propagation(ws_mode):
    """LeakyReLU(0.2) on the load (the stem's activation) and a NaN in x reaching exactly its row's outputs"""
    ops = ws_mode
    M, N, K, G = 256, 64, 64, 1
    x, w, bias, sc, sh = _operands(M, N, K, G, True, seed=5)
    x[17, 3] = float("nan")
    ref = _fp64_forward(x, w, bias, sc, sh, 2, M, N, K, G)
    ops.set_tuning("ws_gemm", 1)
    ops.launch_counters(reset=True)
    out, _ = ops.linear_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), M, N, K, G, sc.to(DEV), sh.to(DEV), ops.ACT_LEAKY, ops.ACT_NONE,
                            want_stat=True)
    assert ops.launch_counters()["ws_fwd"] == 1
    out = out.float().cpu().double()
    assert torch.isnan(out[17]).all() and not torch.isnan(out[:17]).any() and not torch.isnan(out[18:]).any()
    ok = torch.ones(M, dtype=torch.bool)
    ok[17] = False
    assert ((out[ok] - ref[ok]).abs() <= ref[ok].abs() * 2.0 ** -8 + 2e-3).all()


def test_ws_forward_refuses_shapes_outside_its_table(ws_mode):
    """rows that are not whole 128-row tiles and layers whose weights do not fit take the tile kernels (same results as before)"""
    ops = ws_mode
    ops.set_tuning("ws_gemm", 1)
    for (M, N, K, G) in [(192, 64, 64, 1), (256, 512, 128, 1), (256, 96, 64, 1)]:
        x, w, bias, _, _ = _operands(M, N, K, G, False, seed=9)
        ops.launch_counters(reset=True)
        out, _ = ops.linear_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), M, N, K, G, want_stat=True)
        c = ops.launch_counters()
        assert c["ws_fwd"] == 0 and c["gemm_fwd"] == 1
        ref = _fp64_forward(x, w, bias, None, None, 0, M, N, K, G)
        assert ((out.float().cpu().double() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-3).all()


# (M, Nout, K, groups): backward-data launches, din (M, groups * K) = dout (M, groups * Nout) @ W
BWD_SHAPES = [(512, 64, 64, 1), (384, 32, 32, 4), (256, 64, 128, 1), (384, 256, 64, 1), (256, 64, 256, 1), (384, 128, 128, 1),
              (256, 64, 64, 4), (256, 128, 256, 1), (256, 128, 128, 4)]


@pytest.mark.parametrize("M,N,K,G", BWD_SHAPES)
@pytest.mark.parametrize("with_addend,with_bn,act", [(False, False, 0), (True, False, 0), (False, True, 1), (True, True, 0), (True, True, 2)])
def test_ws_backward_data_matches_fp64_and_the_tile_kernel(ws_mode, M, N, K, G, with_addend, with_bn, act):
    """din = addend + dout W in one rounding; with bn = (r, affine, act) the launch also returns the BatchNorm-backward column sums
    sum(g), sum(g * xhat) of the ROUNDED din per 128-row tile (torch_vertex.py:152-162 backward, graph_encoder.py:74-77 backward)"""
    ops = ws_mode
    g = torch.Generator(device="cpu").manual_seed(77 + M + N + K + G + act)
    dout = (torch.randn(M, G * N, generator=g) * 0.7).to(torch.bfloat16)
    w = torch.randn(G * N, K, generator=g) * N ** -0.5
    addend = (torch.randn(M, G * K, generator=g) * 0.5).to(torch.bfloat16) if with_addend else None
    r = (torch.randn(M, G * K, generator=g) * 1.2).to(torch.bfloat16) if with_bn else None
    C = G * K
    scale = (1 + 0.2 * torch.randn(C, generator=g))
    scale[::5] = -scale[::5]
    shift, mean = 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    invstd = 0.5 + torch.rand(C, generator=g)
    wb = w.to(torch.bfloat16).double()
    ref = torch.empty(M, C, dtype=torch.float64)
    for gi in range(G):
        ref[:, gi * K:(gi + 1) * K] = dout[:, gi * N:(gi + 1) * N].double() @ wb[gi * N:(gi + 1) * N]
    if with_addend:
        ref = ref + addend.double()
    dd, wd = dout.to(DEV), w.to(DEV)
    ad = addend.to(DEV) if with_addend else None
    bn = None
    if with_bn:
        aff = ops.BNAffine(scale.to(DEV), shift.to(DEV), mean.to(DEV), invstd.to(DEV))
        bn = (r.to(DEV), aff, act)
    res = {}
    for mode in (0, 2):
        ops.set_tuning("ws_gemm", mode)
        ops.launch_counters(reset=True)
        out = ops.linear_bwd_data(dd, wd, M, N, K, G, addend=ad, bn=bn)
        torch.cuda.synchronize()
        assert ops.launch_counters()["ws_bwd_data"] == (1 if mode else 0)
        din, part = out if with_bn else (out, None)
        res[mode] = (din.float().cpu().double(), None if part is None else part.cpu().double())
    tol = ref.abs() * 2.0 ** -8 + 2e-3
    tiles = M // 128
    for mode in (0, 2):
        din, part = res[mode]
        assert ((din - ref).abs() <= tol).all(), (mode, float(((din - ref).abs() / tol).max()))
        if with_bn:
            # the sums are defined on the kernel's OWN rounded output (what a separate reduce pass would read back)
            z = scale.double() * r.double() + shift.double()
            slope = {0: 1.0, 1: 0.0, 2: 0.2}[act]
            gg = torch.where(z > 0, din, din * slope) if act else din
            xhat = (r.double() - mean.double()) * invstd.double()
            s0 = gg.reshape(tiles, 128, C).sum(1)
            s1 = (gg * xhat).reshape(tiles, 128, C).sum(1)
            a0 = gg.abs().reshape(tiles, 128, C).sum(1)
            a1 = (gg * xhat).abs().reshape(tiles, 128, C).sum(1)
            assert part.shape == (2, tiles, C)
            assert ((part[0] - s0).abs() <= 4e-6 * a0 + 1e-5).all(), float((part[0] - s0).abs().max())
            assert ((part[1] - s1).abs() <= 4e-6 * a1 + 1e-5).all(), float((part[1] - s1).abs().max())
    diff = (res[0][0] != res[2][0]).double().mean()
    assert diff < 2e-2, float(diff)


@pytest.mark.parametrize("M,N,K,G", BWD_SHAPES)
@pytest.mark.parametrize("act,with_addend,with_bn", [(1, False, False), (0, True, True), (2, False, True), (1, True, False)])
def test_ws_backward_data_with_the_batchnorm_backward_on_its_operand_load(ws_mode, M, N, K, G, act, with_addend, with_bn):
    """dr = BN-backward(dy, r) evaluated on the operand load of the weight-stationary backward-data GEMM (one evaluation per row: a
    workgroup owns all output columns), dr written once for the weight gradient, din = addend + dr W: against the unfused pair
    (nsid_bn_bwd_apply + the tile GEMM). dr may differ from the apply pass by one fp32 rounding in front of the bf16 store."""
    ops = ws_mode
    g = torch.Generator(device="cpu").manual_seed(311 + M + N + K + G + act)
    C = G * N
    dy = (torch.randn(M, C, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    r = (torch.randn(M, C, generator=g) * 1.2).to(torch.bfloat16).to(DEV)
    w = (torch.randn(C, K, generator=g) * N ** -0.5).to(DEV)
    addend = (torch.randn(M, G * K, generator=g) * 0.5).to(torch.bfloat16).to(DEV) if with_addend else None
    scale = 1 + 0.2 * torch.randn(C, generator=g)
    scale[::5] = -scale[::5]
    aff = ops.BNAffine(scale.to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV),
                       (0.5 + torch.rand(C, generator=g)).to(DEV))
    bn = None
    if with_bn:
        Ck = G * K
        bn = ((torch.randn(M, Ck, generator=g) * 1.1).to(torch.bfloat16).to(DEV),
              ops.BNAffine((1 + 0.2 * torch.randn(Ck, generator=g)).to(DEV), (0.3 * torch.randn(Ck, generator=g)).to(DEV),
                           (0.2 * torch.randn(Ck, generator=g)).to(DEV), (0.5 + torch.rand(Ck, generator=g)).to(DEV)), 1)
    res = {}
    keep = ops.FUSE_BN_BWD_APPLY
    try:
        for mode, fuse in ((0, 0), (7, 0)):
            ops.set_tuning("ws_gemm", mode)
            ops.FUSE_BN_BWD_APPLY = fuse
            dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            ops.launch_counters(reset=True)
            dr, din, part = ops.bn_backward_linear_bwd_data(dy, r, aff, act, dgamma, dbeta, None, w, M, N, K, G, addend=addend, bn=bn,
                                                            site=ops.SITE_FFN, inplace=False)
            torch.cuda.synchronize()
            c = ops.launch_counters()
            assert c["ws_bwd_bnapply"] == (1 if mode else 0) and c["bn_bwd_apply"] == (0 if mode else 1), c
            res[mode] = (dr.float().cpu().double(), din.float().cpu().double(), None if part is None else part.cpu().double(),
                         dgamma.cpu(), dbeta.cpu())
    finally:
        ops.FUSE_BN_BWD_APPLY = keep
    dr0, din0, p0, dg0, db0 = res[0]
    dr1, din1, p1, dg1, db1 = res[7]
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)                 # the same reduce + finalize launches
    # dr: the apply pass's values except where one fp32 rounding moved a bf16 rounding (an ulp, rarely)
    assert ((dr1 - dr0).abs() <= dr0.abs() * 2.0 ** -7 + 1e-6).all()
    assert float((dr1 != dr0).double().mean()) < 2e-3
    # din from the kernel's OWN dr, in fp64
    wb = w.cpu().to(torch.bfloat16).double()
    ref = torch.empty(M, G * K, dtype=torch.float64)
    for gi in range(G):
        ref[:, gi * K:(gi + 1) * K] = dr1[:, gi * N:(gi + 1) * N] @ wb[gi * N:(gi + 1) * N]
    if with_addend:
        ref = ref + addend.float().cpu().double()
    assert ((din1 - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-3).all()
    assert float((din1 != din0).double().mean()) < 3e-2
    if with_bn:
        assert ((p1 - p0).abs() <= 1e-3 * p0.abs() + 0.3).all()            # sums over values that differ where din does
