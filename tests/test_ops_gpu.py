"""GPU parity of every C-ABI op (through neuralsampleid_amd.ops) against the oracle and the reference goldens.
Tolerances: fp32 GEMM-shaped ops 2e-4 relative to the operand scale (summation order differs from oneDNN);
integer outputs (kNN ids, arg-max) exact outside the recorded near-tie margins."""
import numpy as np
import pytest
import torch

from compare import absmax, maxerr
from conftest import from_rows, to_rows
from oracle import ref_torch as R
from synth import GRAFP_CFG, synth_randn, synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from neuralsampleid_amd import ops as o
    return o


def rnd(tag, *shape):
    return synth_randn("ops:" + tag, *shape)


def close(a, b, tol=2e-4, what=""):
    """b: the reference (a tensor, or a golden tensor in compact form: tests/compare.py)"""
    scale = max(1.0, absmax(b))
    err = maxerr(a, b)
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e})"


def act_ref(x, act):
    return {0: x, 1: torch.relu(x), 2: torch.nn.functional.leaky_relu(x, 0.2), 3: torch.nn.functional.elu(x)}[act]


LIN_CASES = [  # M, Nout, K, groups, affine, act_in, act_out, bias
    (512, 64, 64, 1, False, 0, 0, True),
    (200, 256, 64, 1, True, 1, 0, False),       # ragged M, BN+ReLU on load
    (384, 32, 32, 4, False, 0, 0, True),        # grouped conv of the C=64 stage (narrow group)
    (256, 128, 128, 4, True, 2, 0, True),
    (640, 64, 8, 1, False, 0, 0, False),        # stem: K=8
    (256, 4096, 1024, 1, False, 0, 3, True),    # projector fc1 + ELU
    (130, 192, 384, 1, False, 0, 0, True),      # downsample-like K=3C, ragged M
]


@pytest.mark.parametrize("M,Nout,K,groups,affine,act_in,act_out,has_bias", LIN_CASES)
def test_linear_fwd(ops, M, Nout, K, groups, affine, act_in, act_out, has_bias):
    x = rnd(f"x{M}{K}{groups}", M, groups * K)
    w = rnd(f"w{Nout}{K}{groups}", groups * Nout, K) * K ** -0.5
    bias = rnd(f"b{Nout}", groups * Nout) if has_bias else None
    sc = 1 + 0.2 * rnd("sc", groups * K) if affine else None
    sh = 0.3 * rnd("sh", groups * K) if affine else None
    xin = act_ref(x * sc + sh, act_in) if affine else x
    ref = torch.cat([xin[:, g * K:(g + 1) * K].double() @ w[g * Nout:(g + 1) * Nout].double().t()
                     for g in range(groups)], dim=1)
    if has_bias:
        ref = ref + bias.double()
    d = lambda t: None if t is None else t.to(DEV)
    out, stat = ops.linear_fwd(d(x), d(w), d(bias), M, Nout, K, groups, d(sc), d(sh), act_in, act_out, want_stat=True)
    close(out, act_ref(ref, act_out), what="out")
    tiles = ops.row_tiles(M)
    pad = torch.zeros(tiles * 128 - M, groups * Nout, dtype=torch.float64)
    rt = torch.cat([ref, pad]).reshape(tiles, 128, -1)
    close(stat[0], rt.sum(1), tol=5e-4, what="stat sum")
    close(stat[1], (rt * rt).sum(1), tol=5e-4, what="stat sumsq")


def test_linear_fwd_ksplit(ops):
    M, Nout, K = 256, 128, 4096
    x, w, b = rnd("ksx", M, K), rnd("ksw", Nout, K) * K ** -0.5, rnd("ksb", Nout)
    out, _ = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), M, Nout, K, ksplit=8)
    close(out, x.double() @ w.double().t() + b.double(), what="ksplit")
    # with the ELU of the projector's first linear (simclr/simclr.py:25-28): its own pass over the finished sums
    M, Nout, K = 256, 4096, 1024
    x, w, b = rnd("ksx2", M, K), rnd("ksw2", Nout, K) * K ** -0.5, rnd("ksb2", Nout)
    for ks in (1, 2, 4):
        out, _ = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), M, Nout, K, act_out=3, ksplit=ks)
        close(out, torch.nn.functional.elu(x.double() @ w.double().t() + b.double()), what=f"ksplit {ks} + ELU")


@pytest.mark.parametrize("M,Nout,K,groups,add", [(512, 64, 64, 1, True), (200, 256, 64, 1, False),
                                                 (384, 32, 32, 4, False), (256, 128, 512, 1, True),
                                                 (256, 128, 4096, 1, False)])
def test_linear_bwd_data(ops, M, Nout, K, groups, add):
    dout = rnd(f"bd{M}{Nout}", M, groups * Nout)
    w = rnd(f"bw{Nout}{K}", groups * Nout, K) * Nout ** -0.5
    addend = rnd("badd", M, groups * K) if add else None
    ref = torch.cat([dout[:, g * Nout:(g + 1) * Nout].double() @ w[g * Nout:(g + 1) * Nout].double()
                     for g in range(groups)], dim=1)
    if add:
        ref = ref + addend.double()
    got = ops.linear_bwd_data(dout.to(DEV), w.to(DEV), M, Nout, K, groups, None if addend is None else addend.to(DEV))
    close(got, ref, what="din")


@pytest.mark.parametrize("M,Nout,K", [(256, 4096, 1024), (256, 1024, 512), (200, 2048, 192), (8, 4096, 1024)])
def test_linear_bwd_data_split_reduction(ops, M, Nout, K):
    """the projector head's backward-data (simclr/simclr.py:25-28 backward; B rows, Nout up to 4096): a handful of tiles with a long
    reduction is split over up to 256 workgroups (fp32 atomics into the output, which the launch zeroes itself), against fp64 and against the unsplit launch"""
    dout = rnd(f"sd{M}{Nout}", M, Nout)
    w = rnd(f"sw{Nout}{K}", Nout, K) * Nout ** -0.5
    ref = dout.double() @ w.double()
    out = torch.full((M, K), 7.0, device=DEV)                 # stale contents: the launch must not rely on a zeroed buffer
    ops.launch_counters(reset=True)
    got = ops.linear_bwd_data(dout.to(DEV), w.to(DEV), M, Nout, K, 1, out=out)
    torch.cuda.synchronize()
    assert ops.launch_counters()["gemm_bwd_split"] == 1
    close(got, ref, what="din (split)")
    try:
        ops.set_tuning("bwd_split_max_tiles", 0)
        ops.launch_counters(reset=True)
        plain = ops.linear_bwd_data(dout.to(DEV), w.to(DEV), M, Nout, K, 1)
        assert ops.launch_counters()["gemm_bwd_split"] == 0
    finally:
        ops.reset_tuning()
    close(got, plain, tol=2e-5, what="split vs unsplit")


@pytest.mark.parametrize("M,Nout,K,groups,affine,act", [(1024, 64, 64, 1, False, 0), (700, 256, 64, 1, True, 1),
                                                        (512, 32, 32, 4, False, 0), (4096, 128, 512, 1, True, 1),
                                                        (256, 4096, 1024, 1, False, 0), (2048, 64, 8, 1, False, 0)])
def test_linear_bwd_weight(ops, M, Nout, K, groups, affine, act):
    dout = rnd(f"wd{M}{Nout}", M, groups * Nout)
    x = rnd(f"wx{M}{K}", M, groups * K)
    sc = 1 + 0.2 * rnd("wsc", groups * K) if affine else None
    sh = 0.3 * rnd("wsh", groups * K) if affine else None
    xin = act_ref(x * sc + sh, act) if affine else x
    ref = torch.cat([dout[:, g * Nout:(g + 1) * Nout].double().t() @ xin[:, g * K:(g + 1) * K].double()
                     for g in range(groups)], dim=0)
    d = lambda t: None if t is None else t.to(DEV)
    dw = torch.ones(groups * Nout, K, device=DEV)        # "+=": starts from ones
    ops.linear_bwd_weight(d(dout), d(x), dw, M, Nout, K, groups, d(sc), d(sh), act)
    close(dw, ref + 1.0, tol=3e-4, what="dw")


def test_colsum(ops):
    x = rnd("cs", 700, 256)
    out = torch.ones(256, device=DEV)
    ops.colsum_acc(x.to(DEV), out)
    close(out, x.double().sum(0) + 1, what="colsum")


@pytest.mark.parametrize("M,C,act", [(512, 64, 0), (300, 256, 1), (1024, 2048, 1), (256, 80, 2)])
def test_batchnorm_train(ops, M, C, act):
    r = (rnd(f"bnr{M}{C}", M, C) * 1.5 + 0.7)
    gamma, beta = 1 + 0.1 * rnd("bng", C), 0.1 * rnd("bnb", C)
    rm, rv = 0.1 * rnd("bnrm", C), 0.5 + rnd("bnrv", C).abs()
    # statistics partials come from the GEMM epilogue: use an identity-free path (x @ I) to get them
    eye = torch.eye(C)
    out, stat = ops.linear_fwd(r.to(DEV), eye.to(DEV), None, M, C, C, want_stat=True)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    aff = ops.bn_finalize(stat, M, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, nbt)
    # reference: torch BatchNorm semantics in fp64
    x64 = r.double()
    mean, var = x64.mean(0), x64.var(0, unbiased=False)
    close(aff.mean, mean, tol=1e-6, what="mean")
    close(aff.invstd, 1 / torch.sqrt(var + 1e-5), tol=1e-5, what="invstd")
    close(rm_d, 0.9 * rm.double() + 0.1 * mean, tol=1e-6, what="running_mean")
    close(rv_d, 0.9 * rv.double() + 0.1 * var * M / (M - 1), tol=1e-6, what="running_var")
    assert int(nbt) == 1
    res = rnd("bnres", M, C)
    y = ops.bn_apply(out, aff, act, res.to(DEV))
    xg = r.clone().double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yref = act_ref((xg - xg.mean(0)) / torch.sqrt(xg.var(0, unbiased=False) + 1e-5) * g64 + b64, act)
    close(y, yref + res.double(), tol=2e-5, what="bn_apply")
    dout = rnd("bndo", M, C)
    yref.backward(dout.double())
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dr = ops.bn_backward(dout.to(DEV), out, aff, act, dgamma, dbeta)
    close(dr, xg.grad, tol=2e-5, what="dr")
    close(dgamma, g64.grad, tol=2e-5, what="dgamma")
    close(dbeta, b64.grad, tol=2e-5, what="dbeta")
    ev = ops.bn_eval_affine(gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV))
    close(ops.bn_apply(out, ev), (x64 - rm.double()) / torch.sqrt(rv.double() + 1e-5) * gamma.double() + beta.double(),
          tol=2e-5, what="eval affine")


def knn_mismatch(idx, gold_idx, gap, tol=1e-4):
    a = np.sort(idx.cpu().numpy(), axis=-1)
    b = np.sort(np.asarray(gold_idx), axis=-1)
    bad = (a != b).any(axis=-1)
    return int((bad & (np.asarray(gap) >= tol)).sum()), int(bad.sum())


def test_deferred_running_statistics_equal_two_ordered_finalizes(ops):
    """nsid_bn_finalize_deferred x2 + ONE nsid_bn_running_update = nsid_bn_finalize for view i, then for view j: same
    affine per view, same running statistics bit for bit, num_batches_tracked += 2 (reference order: simclr.py:36,42)"""
    M, C = 640, 96
    xs = [(rnd(f"dv{v}", M, C) * (1.0 + v) + 0.3 * v).to(DEV) for v in range(2)]
    w = (rnd("dw", C, C) * C ** -0.5).to(DEV)
    gamma, beta = (1 + 0.1 * rnd("dg", C)).to(DEV), (0.1 * rnd("db", C)).to(DEV)
    stats = [ops.linear_fwd(x, w, None, M, C, C, want_stat=True)[1] for x in xs]
    rm0, rv0 = rnd("drm", C).to(DEV), (0.5 + rnd("drv", C).abs()).to(DEV)
    # ordered: two finalize launches that each update the running statistics
    rm_a, rv_a, nbt_a = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    affs = [ops.bn_finalize(st, M, gamma, beta, rm_a, rv_a, nbt_a) for st in stats]
    # deferred: statistics only, then one update launch for both views
    rm_b, rv_b, nbt_b = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    pend = []
    for st, ref in zip(stats, affs):
        aff, uvar = ops.bn_finalize_deferred(st, M, gamma, beta)
        for name in ("scale", "shift", "mean", "invstd"):
            assert torch.equal(getattr(aff, name), getattr(ref, name)), name
        pend.append((rm_b, rv_b, nbt_b, aff.mean, uvar))
    ops.bn_running_update([pend[0]], [pend[1]])
    assert torch.equal(rm_b, rm_a) and torch.equal(rv_b, rv_a) and int(nbt_b) == int(nbt_a) == 2
    # one view only
    rm_c, rv_c, nbt_c = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    rm_d, rv_d, nbt_d = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    ops.bn_finalize(stats[0], M, gamma, beta, rm_c, rv_c, nbt_c)
    aff, uvar = ops.bn_finalize_deferred(stats[0], M, gamma, beta)
    ops.bn_running_update([(rm_d, rv_d, nbt_d, aff.mean, uvar)] * 1)
    assert torch.equal(rm_d, rm_c) and torch.equal(rv_d, rv_c) and int(nbt_d) == 1
    # more layers than one launch carries (16)
    many = [(rm0.clone(), rv0.clone(), None, aff.mean, uvar) for _ in range(19)]
    ops.bn_running_update(many)
    assert all(torch.equal(m[0], rm_c) and torch.equal(m[1], rv_c) for m in many)


def test_batchnorm_eval_affine_is_cached_until_a_tensor_changes(ops):
    """eval-mode scale/shift are computed once per layer and follow in-place updates of any of the four tensors"""
    C = 96
    g, b = (1 + 0.1 * synth_randn("eg", C)).to(DEV), (0.1 * synth_randn("eb", C)).to(DEV)
    rm, rv = synth_randn("erm", C).to(DEV), (0.5 + synth_randn("erv", C).abs()).to(DEV)
    ref = lambda: (g / torch.sqrt(rv + 1e-5), b - rm * g / torch.sqrt(rv + 1e-5))
    a1 = ops.bn_eval_affine(g, b, rm, rv)
    assert torch.allclose(a1.scale, ref()[0], rtol=1e-6, atol=1e-7) and torch.allclose(a1.shift, ref()[1], rtol=1e-6, atol=1e-6)
    a2 = ops.bn_eval_affine(g, b, rm, rv)
    assert a2.scale.data_ptr() == a1.scale.data_ptr()                 # served from the cache
    rv.mul_(2.0)                                                      # e.g. load_state_dict / a training step
    a3 = ops.bn_eval_affine(g, b, rm, rv)
    assert a3.scale.data_ptr() != a1.scale.data_ptr()
    assert torch.allclose(a3.scale, ref()[0], rtol=1e-6, atol=1e-7) and torch.allclose(a3.shift, ref()[1], rtol=1e-6, atol=1e-6)


# k = 18 entries: BASELINE config 4's four shapes — (C64,N256,d1) / (C128,N128,d2) -> knn_sel_kernel, (C256,N64,d3) / (C512,N32,d1)
# -> knn_rank_kernel; c80n256 (size 's' channel count, not a power of two) -> the general strip kernel knn_kernel
@pytest.mark.parametrize("tag,kds", [("c64n256", [(3, 1), (5, 1), (4, 2), (18, 3), (18, 1)]),
                                     ("c128n128", [(3, 1), (18, 2)]),
                                     ("c256n64", [(3, 1), (18, 3), (9, 2)]),
                                     ("c512n32", [(3, 1), (5, 2), (18, 1)]),
                                     ("c80n256", [(3, 1), (9, 2)])])
def test_knn_graph_golden(ops, golden, tag, kds):
    g = golden("knn_" + tag)
    y = to_rows(g.t("x"))
    B, N, C = y.shape
    rows = y.reshape(B * N, C).contiguous().to(DEV)
    for k, d in kds:
        idx = ops.knn_graph(rows, B, N, C, k, d)
        gap = g[f"mingap_k{k}_d{d}"] if d > 1 else g[f"setgap_k{k}_d{d}"]
        hard, soft = knn_mismatch(idx, g[f"idx_k{k}_d{d}"], gap)
        assert hard == 0, (k, d, hard, soft)
        assert soft <= max(2, B * N // 200), (k, d, soft)
        assert (idx[..., 0].cpu() == torch.arange(N)).all()
        # outside near-ties the ORDER matches too
        exact = (idx.cpu().numpy() == g[f"idx_k{k}_d{d}"]).all(-1) | (g[f"mingap_k{k}_d{d}"] < 1e-5)
        assert exact.mean() > 0.98, (k, d, exact.mean())


@pytest.mark.parametrize("N,C,k,d", [(128, 128, 18, 2), (64, 256, 18, 3), (256, 64, 18, 1), (256, 64, 3, 1), (32, 512, 18, 1)])
def test_knn_graph_duplicated_nodes_take_the_lower_index(ops, N, C, k, d):
    """exact ties (duplicated nodes: identical rows give bit-identical distances) in every selection kernel: rank counting with the
    row-sum tie detection (N = 128), the one-pass form (N <= 64), threshold select (N = 256, k*d > 8) and the register lists
    (k*d <= 8) order equal distances by node index, like a stable sort of the reference's distances"""
    B = 3
    x = rnd(f"dupknn{N}", B * N, C)
    x = x.reshape(B, N, C).clone()
    x[0, 5] = x[0, 3]                       # a pair
    x[1, N - 1] = x[1, 2]
    x[1, N // 2] = x[1, 2]                  # a triple
    x[2, 1] = x[2, 0]
    x[2, 7] = x[2, 6]                       # two pairs in one clip
    idx = ops.knn_graph(x.reshape(B * N, C).contiguous().to(DEV), B, N, C, k, d).cpu()
    xn = torch.nn.functional.normalize(x.double(), dim=-1)
    dist = (xn * xn).sum(-1, keepdim=True) - 2 * xn @ xn.transpose(1, 2) + (xn * xn).sum(-1).unsqueeze(1)
    for (b, i, twins) in [(0, 3, [3, 5]), (0, 5, [3, 5]), (1, 2, [2, N // 2, N - 1]), (1, N - 1, [2, N // 2, N - 1]), (2, 0, [0, 1]),
                          (2, 6, [6, 7])]:
        got = idx[b, i].tolist()
        # the duplicates of node i are at distance 0 from it, bit-identical: they must lead the list in index order (dilation keeps
        # every d-th of the sorted neighbours)
        want = twins[::d][:k]
        assert got[:len(want)] == want, (N, b, i, got[:4], want)
    # everything else still matches a stable sort of fp64 distances wherever those are well separated
    order = torch.argsort(dist, dim=-1, stable=True)[..., :k * d:d]
    srt = torch.sort(dist, dim=-1).values
    gap = (srt[..., 1:k * d + 1] - srt[..., :k * d]).clamp_min(0)
    clear = (gap[..., 2:].min(-1).values > 1e-4)              # (the first gaps are the exact ties made above)
    same = (idx.long() == order).all(-1) | ~clear
    assert same.float().mean() > 0.97, float(same.float().mean())


def test_knn_graph_affine(ops):
    B, N, C, k = 4, 64, 256, 5
    r = rnd("knnaff", B * N, C)
    sc, sh = 1 + 0.3 * rnd("knnsc", C), 0.5 * rnd("knnsh", C)
    aff = ops.BNAffine(sc.to(DEV), sh.to(DEV))
    idx = ops.knn_graph(r.to(DEV), B, N, C, k, 1, aff)
    ref = R._knn_graph((r * sc + sh).reshape(B, N, C), k, 1)
    same = (np.sort(idx.cpu().numpy(), -1) == np.sort(ref.numpy(), -1)).all(-1)
    assert same.mean() > 0.995


@pytest.mark.parametrize("N,C", [(256, 64), (128, 128), (64, 256), (32, 512)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_knn2_pair_variant_equals_the_default(ops, N, C, dt):
    """knn2_pair_kernel (128 VGPRs, two workgroups per CU, no fragment prefetch; picked from `knn_pair_min` clips per launch)
    computes the same distances in the same order as knn2_kernel: identical ids for every k*d of the fast path, and the launch
    counter says which one ran"""
    B = 24
    r = rnd(f"knnpair{N}", B * N, C).to(DEV).to(dt)
    aff = ops.BNAffine(torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1)
    try:
        for k, d in ((3, 1), (5, 1), (4, 2), (2, 3)):
            ops.set_tuning("knn_pair_min", 0)
            want = ops.knn_graph(r, B, N, C, k, d, aff)
            ops.set_tuning("knn_pair_min", 1)
            c0 = ops.launch_counters()["knn2_pair"]
            got = ops.knn_graph(r, B, N, C, k, d, aff)
            assert ops.launch_counters()["knn2_pair"] == c0 + 1
            assert (got == want).all(), (k, d)
    finally:
        ops.reset_tuning()


@pytest.mark.parametrize("N,C,k", [(256, 64, 3), (256, 64, 8), (128, 128, 3), (64, 256, 3), (32, 512, 3), (128, 128, 18), (64, 256, 18)])
def test_knn_split_fp16_distance_error_bound(ops, N, C, k):
    """every fast kNN path computes the distance products from a two-part fp16 split (csrc/knn.hip); its error against an
    fp64 evaluation is ~3e-7 on unit-norm rows, PROVIDED the matrix core keeps fp16 subnormal inputs (the low parts of features around
    0.04 are subnormal: flushed, the error grows to ~1e-5). The kernels do not export distances, so the bound is stated on what they
    decide: on every row whose first k+1 fp64 distances are separated by more than 2e-6, the neighbour ids must be the fp64 ranking."""
    B = 48
    r = rnd(f"knnsplit{N}{C}", B * N, C).to(DEV)
    idx = ops.knn_graph(r, B, N, C, k, 1, None).long().cpu()
    y = r.double().reshape(B, N, C).cpu()
    y = y / y.norm(dim=2, keepdim=True).clamp_min(1e-12)
    sq = (y * y).sum(2)
    D = sq[:, :, None] - 2.0 * torch.bmm(y, y.transpose(1, 2)) + sq[:, None, :]
    Ds, order = torch.sort(D, dim=2, stable=True)
    gaps = (Ds[:, :, 1:k + 1] - Ds[:, :, :k]).min(dim=2).values           # separation inside the first k+1 (incl. the k / k+1 boundary)
    clear = gaps > 2e-6
    assert float(clear.float().mean()) > 0.5, "the input has too many near-ties for this test to say anything"
    same = (idx == order[:, :, :k]).all(dim=2)
    bad = int((clear & ~same).sum())
    assert bad == 0, f"{bad} rows with fp64 gaps > 2e-6 got other neighbours: the split-fp16 distances are off by more than 1e-6"


@pytest.mark.parametrize("N,C,k,d", [(128, 128, 18, 2), (64, 256, 18, 3), (32, 512, 18, 1), (128, 128, 16, 4)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_knn_threshold_select_equals_rank_counting(ops, N, C, k, d, dt):
    """the two large-k*d kernels compute the same distances in the same order and break ties the same way (lower index): on low-rank
    features with duplicated nodes (many exact ties) knn_sel_kernel<T, N/16> and knn_rank_kernel must return identical ids"""
    B = 12
    g = torch.Generator().manual_seed(N + k)
    r = (torch.randn(B * N, 6, generator=g) @ torch.randn(6, C, generator=g)).to(DEV).to(dt)
    r[:N * 2:7] = r[1:N * 2 + 1:7].clone()
    aff = ops.BNAffine(torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV) * 0.1)
    try:
        ops.set_tuning("knn_sel_min_n", 512)
        c0 = ops.launch_counters()
        want = ops.knn_graph(r, B, N, C, k, d, aff)
        ops.set_tuning("knn_sel_min_n", 32)
        got = ops.knn_graph(r, B, N, C, k, d, aff)
        c1 = ops.launch_counters()
        assert c1["knn_rank"] == c0["knn_rank"] + 1 and c1["knn_sel"] == c0["knn_sel"] + 1
        assert (got == want).all()
    finally:
        ops.reset_tuning()


def test_mr_aggregate_golden(ops, golden):
    g = golden("mragg_c64n256")
    y = to_rows(g.t("x"))
    B, N, C = y.shape
    idx = g.t("idx").to(DEV)
    rows = y.reshape(B * N, C).contiguous().to(DEV)
    u, amax = ops.mr_aggregate_fwd(rows, idx, B, N, C)
    assert torch.equal(from_rows(u.cpu().reshape(B, N, 2 * C)), g.t("u"))          # exact: gather, subtract, max
    gu = to_rows(g.t("gu")).reshape(B * N, 2 * C).contiguous().to(DEV)
    dy = ops.mr_aggregate_bwd(gu, idx, amax, B, N, C)
    close(from_rows(dy.cpu().reshape(B, N, C)), g.t("dx"), tol=1e-6, what="mr dx")


def test_mr_aggregate_affine_k18(ops):
    B, N, C, k = 3, 128, 128, 18
    r = rnd("mraff", B * N, C)
    sc, sh = 1 + 0.3 * rnd("mrsc", C), 0.5 * rnd("mrsh", C)
    y = (r * sc + sh).reshape(B, N, C).requires_grad_(True)
    idx = R._knn_graph(y.detach(), k, 2)
    uref = R.mr_aggregate(y, idx)
    gu = rnd("mrgu", B, N, 2 * C)
    (uref * gu).sum().backward()
    aff = ops.BNAffine(sc.to(DEV), sh.to(DEV))
    idx_d = idx.to(torch.int32).to(DEV)
    u, amax = ops.mr_aggregate_fwd(r.to(DEV), idx_d, B, N, C, aff)
    close(u, uref.reshape(B * N, 2 * C), tol=1e-6, what="u")
    dy = ops.mr_aggregate_bwd(gu.reshape(B * N, 2 * C).to(DEV), idx_d, amax, B, N, C)
    close(dy, y.grad.reshape(B * N, C), tol=1e-5, what="dy")


def test_downsample_pieces(ops):
    B, N, C, Co = 3, 64, 64, 128
    x = rnd("dsx", B, N, C)
    w = rnd("dsw", Co, C, 3, 3) * (3 * C) ** -0.5
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1))
    No = ops.ds_out_nodes(N)
    colref = torch.cat([xp[:, t:t + 2 * No:2, :] for t in range(3)], dim=-1).reshape(B * No, 3 * C)
    col = ops.im2col3_fwd(x.reshape(B * N, C).to(DEV), B, N, C)
    assert torch.equal(col.cpu(), colref)
    wp = ops.pack_ds_weight(w.to(DEV))
    assert torch.equal(wp.cpu(), w[:, :, :, 1].permute(0, 2, 1).reshape(Co, 3 * C))
    dcol = rnd("dsdcol", B * No, 3 * C)
    xg = x.clone().requires_grad_(True)
    xpg = torch.nn.functional.pad(xg, (0, 0, 1, 1))
    (torch.cat([xpg[:, t:t + 2 * No:2, :] for t in range(3)], dim=-1).reshape(B * No, 3 * C) * dcol).sum().backward()
    close(ops.im2col3_bwd(dcol.to(DEV), B, N, C), xg.grad.reshape(B * N, C), tol=1e-6, what="col2im")
    dwp = rnd("dsdwp", Co, 3 * C)
    dw = torch.ones(Co, C, 3, 3, device=DEV)
    ops.unpack_ds_wgrad(dwp.to(DEV), dw)
    ref = torch.ones(Co, C, 3, 3)
    ref[:, :, :, 1] += dwp.reshape(Co, 3, C).permute(0, 2, 1)
    assert torch.equal(dw.cpu(), ref)


@pytest.mark.parametrize("B,N,C,Co,dtype", [(3, 64, 64, 128, "fp32"), (2, 256, 64, 128, "fp32"), (5, 32, 256, 512, "fp32"),
                                             (3, 64, 64, 128, "bf16"), (4, 128, 128, 256, "bf16"), (256, 64, 256, 512, "bf16")])
def test_downsample3_strided_view_equals_reference_conv(ops, B, N, C, Co, dtype):
    """nsid_downsample3_fwd / _bwd_weight / _bwd_data (GEMMs over a zero-padded strided view of x, no im2col) against
    torch's Conv2d 3x3 stride 2 pad 1 on the (B, C, N, 1) map (encoder/graph_encoder.py:44) in fp64, forward and backward;
    the clip boundaries (left padding of node 0, no right neighbour for the last output node) are where a view could leak"""
    from neuralsampleid_amd import functional as F_
    bf = dtype == "bf16"
    ops.set_gemm_precision(dtype)
    try:
        adt = torch.bfloat16 if bf else torch.float32
        x = rnd(f"d3x{B}{N}{C}", B * N, C).to(adt)
        w = rnd(f"d3w{C}{Co}", Co, C, 3, 3) * (3 * C) ** -0.5
        bias = rnd(f"d3b{Co}", Co)
        dout = rnd(f"d3d{B}{N}{Co}", B * (N // 2), Co).to(adt)
        q = (lambda t: t.to(torch.bfloat16).double()) if bf else (lambda t: t.double())
        x4 = x.double().reshape(B, N, C).transpose(1, 2).unsqueeze(-1).clone().requires_grad_(True)     # (B, C, N, 1)
        w64 = q(w).requires_grad_(True)
        ref = torch.nn.functional.conv2d(x4, w64, bias.double(), stride=2, padding=1)                  # (B, Co, N/2, 1)
        ref.backward(dout.double().reshape(B, N // 2, Co).transpose(1, 2).unsqueeze(-1))
        ref_rows = ref.detach().squeeze(-1).transpose(1, 2).reshape(B * (N // 2), Co)
        wp = ops.pack_ds_weight(w.to(DEV))
        out, stat = ops.downsample3_fwd(x.to(DEV), B, N, C, wp, bias.to(DEV), Co, want_stat=True)
        tol = 2.5e-3 if bf else 2e-4
        assert out.dtype == adt
        err = float((out.double().cpu() - ref_rows).norm() / ref_rows.norm())
        assert err < tol, err
        tiles = ops.row_tiles(B * (N // 2))
        pad = torch.zeros(tiles * 128 - ref_rows.shape[0], Co, dtype=torch.float64)
        close(stat[0], torch.cat([ref_rows, pad]).reshape(tiles, 128, Co).sum(1), tol=1e-3 if bf else 5e-4, what="stat")
        # the im2col form computes the same thing
        col = ops.im2col3_fwd(x.to(DEV), B, N, C)
        out2, _ = ops.linear_fwd(col, wp, bias.to(DEV), B * (N // 2), Co, 3 * C)
        assert float((out.double() - out2.double()).norm() / out2.double().norm()) < (4e-3 if bf else 1e-5)
        # weight gradient (fp32 accumulation; only kernel column 1 receives any)
        dwp = torch.zeros(Co, 3 * C, device=DEV)
        ops.downsample3_bwd_weight(dout.to(DEV), x.to(DEV), dwp, B, N, C, Co)
        gw = w64.grad[:, :, :, 1].permute(0, 2, 1).reshape(Co, 3 * C)
        assert float((dwp.double().cpu() - gw).norm() / gw.norm()) < (1e-4 if bf else 3e-4)
        assert float(w64.grad[:, :, :, 0].abs().max()) == 0.0 and float(w64.grad[:, :, :, 2].abs().max()) == 0.0
        # input gradient
        dx = ops.downsample3_bwd_data(dout.to(DEV), wp, ops.pack_ds_weight_bwd(w.to(DEV)), B, N, C, Co)
        gx = x4.grad.squeeze(-1).transpose(1, 2).reshape(B * N, C)
        assert dx.dtype == adt and float((dx.double().cpu() - gx).norm() / gx.norm()) < tol
        first = torch.arange(B) * N                         # node 0 of every clip: only tap 1 of output node 0 reaches it
        assert float((dx.double().cpu()[first] - gx[first]).abs().max()) < tol * float(gx.abs().max()) * 4
    finally:
        ops.set_gemm_precision("fp32")


def test_peak_patchify_golden(ops, golden):
    g = golden("peak_b8")
    w = synth_tensor("peak_extractor.convs.0.weight", torch.empty(8, 3, 4, 8))
    b = synth_tensor("peak_extractor.convs.0.bias", torch.empty(8))
    spec = g.t("x").to(DEV)
    out, minmax = ops.peak_patchify_fwd(spec, w.to(DEV), b.to(DEV), 4, 8)
    B = spec.shape[0]
    close(out.reshape(B, 256, 8).transpose(1, 2), g.t("y"), tol=1e-5, what="patchify")
    dw, db = torch.zeros(8, 3, 4, 8, device=DEV), torch.zeros(8, device=DEV)
    dout = g.t("gout").transpose(1, 2).reshape(B * 256, 8).contiguous().to(DEV)
    for ws in (True, False):        # the workspace form (per-clip partial sums + a reduce launch) and the atomics form
        dw.zero_(); db.zero_()
        ops.PATCHIFY_BWD_WS = ws
        try:
            ops.peak_patchify_bwd(spec, minmax, out, dout, 4, 8, dw, db)
        finally:
            ops.PATCHIFY_BWD_WS = True
        close(dw, g.t("dweight"), tol=1e-4, what=f"dweight (ws={ws})")
        close(db, g.t("dbias"), tol=1e-4, what=f"dbias (ws={ws})")


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_peak_patchify_backward_workspace_form_at_the_timed_batch(ops, dt):
    """peak_extractor.py:45-70 backward at B = 256 in both storage types: the workspace form accumulates on top of what dw / dbias hold
    (the second view adds to the first) and equals the atomics form to fp32 summation noise"""
    B = 256
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    w, b = rnd("ppw", 8, 3, 4, 8).to(DEV), rnd("ppb", 8).to(DEV)
    spec = (rnd("pps", B, 64, 128) * 20 - 40).to(DEV)
    out, minmax = ops.peak_patchify_fwd(spec, w, b, 4, 8, tdt)
    dout = rnd("ppd", B * 256, 8).to(tdt).to(DEV)
    res = {}
    for ws in (True, False):
        dw, db = torch.ones(8, 3, 4, 8, device=DEV), torch.full((8,), 2.0, device=DEV)
        ops.PATCHIFY_BWD_WS = ws
        try:
            ops.peak_patchify_bwd(spec, minmax, out, dout, 4, 8, dw, db)
        finally:
            ops.PATCHIFY_BWD_WS = True
        res[ws] = (dw.cpu().double(), db.cpu().double())
    scale = max(1.0, float(res[False][0].abs().max()))
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-5 * scale
    assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-5 * max(1.0, float(res[False][1].abs().max()))


def test_head_pieces(ops):
    B, N, C = 5, 32, 512
    x = rnd("hm", B * N, C)
    close(ops.node_mean_fwd(x.to(DEV), B, N, C), x.reshape(B, N, C).double().mean(1), tol=1e-6, what="mean")
    dm = rnd("hdm", B, C)
    close(ops.node_mean_bwd(dm.to(DEV), B, N, C), (dm / N).repeat_interleave(N, 0), tol=1e-6, what="mean bwd")
    pre = rnd("elupre", 64, 4096)
    out = torch.nn.functional.elu(pre)
    dout = rnd("eludo", 64, 4096)
    pg = pre.clone().requires_grad_(True)
    torch.nn.functional.elu(pg).backward(dout)
    close(ops.elu_bwd(dout.to(DEV), out.to(DEV)), pg.grad, tol=1e-6, what="elu bwd")
    p = rnd("l2p", 37, 128).requires_grad_(True)
    zref = torch.nn.functional.normalize(p, p=2, eps=1e-10)
    dz = rnd("l2dz", 37, 128)
    zref.backward(dz)
    z, norm = ops.l2norm_fwd(p.detach().to(DEV), 1e-10)
    close(z, zref, tol=1e-6, what="l2norm")
    close(ops.l2norm_bwd(dz.to(DEV), z, norm, 1e-10), p.grad, tol=1e-6, what="l2norm bwd")


@pytest.mark.parametrize("B", [2, 8, 256])
def test_ntxent_golden(ops, golden, B):
    g = golden(f"ntxent_b{B}")
    zi, zj = g.t("z_i").to(DEV), g.t("z_j").to(DEV)
    loss, dzi, dzj = ops.ntxent_fwd_bwd(zi, zj, float(g["tau"]))
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 2e-6 * max(1.0, abs(float(g["loss"][0])))
    close(dzi, g.t("dz_i"), tol=2e-6, what="dz_i")
    close(dzj, g.t("dz_j"), tol=2e-6, what="dz_j")


def test_ntxent_sharded(ops, golden):
    """two 'ranks' each owning half of the pairs reproduce the global loss and gradients"""
    g = golden("ntxent_b256")
    zi, zj = g.t("z_i").to(DEV), g.t("z_j").to(DEV)
    tau = float(g["tau"])
    parts = [ops.ntxent_fwd_bwd(zi, zj, tau, p0, 128) for p0 in (0, 128)]
    assert abs(float(parts[0][0]) + float(parts[1][0]) - float(g["loss"][0])) < 2e-6
    close(torch.cat([parts[0][1], parts[1][1]]), g.t("dz_i"), tol=2e-6, what="dz_i sharded")
    close(torch.cat([parts[0][2], parts[1][2]]), g.t("dz_j"), tol=2e-6, what="dz_j sharded")
    # ragged: 37 pairs, d = 64
    zi2 = torch.nn.functional.normalize(rnd("ntxr_i", 37, 64), dim=1).requires_grad_(True)
    zj2 = torch.nn.functional.normalize(rnd("ntxr_j", 37, 64), dim=1).requires_grad_(True)
    lref = R.ntxent(zi2, zj2, 0.1)
    lref.backward()
    loss, dzi, dzj = ops.ntxent_fwd_bwd(zi2.detach().to(DEV), zj2.detach().to(DEV), 0.1)
    assert abs(float(loss.detach()) - float(lref.detach())) < 2e-6
    close(dzi, zi2.grad, tol=2e-6, what="ragged dz_i")
    close(dzj, zj2.grad, tol=2e-6, what="ragged dz_j")


def test_ntxent_global_batch_2048_row_slices(ops):
    """BASELINE config 3 on one GPU: the rank-local NT-Xent of each of the 8 ranks of a global batch of 2048 pairs (every
    rank owns 256 pairs = 512 rows of the interleaved 4096 x 4096 problem, all columns) against the oracle's row-sliced
    closed form + autograd; the 8 losses sum to the global mean loss and the concatenated dz equal its gradient
    (reference: ntxent.py:5-30 on the DataParallel-gathered batch, train.py:63)."""
    Bg, d, tau, world = 2048, 128, 0.05, 8
    zi = torch.nn.functional.normalize(rnd("g2048_i", Bg, d), dim=1)
    zj = torch.nn.functional.normalize(zi + 0.5 * rnd("g2048_j", Bg, d), dim=1)
    zi_r, zj_r = zi.clone().requires_grad_(True), zj.clone().requires_grad_(True)
    full = R.ntxent(zi_r, zj_r, tau)
    full.backward()
    zi_d, zj_d = zi.to(DEV), zj.to(DEV)
    z_all = torch.stack((zi, zj), 1).reshape(2 * Bg, d)
    losses, dzi, dzj = [], [], []
    per = Bg // world
    for r in range(world):
        loss, gi, gj = ops.ntxent_fwd_bwd(zi_d, zj_d, tau, r * per, per)
        want = float(R.ntxent_rows(z_all, 2 * r * per, 2 * per, tau)) / (2 * Bg)
        assert abs(float(loss.detach()) - want) < 2e-6 * max(1.0, abs(want)), (r, float(loss.detach()), want)
        losses.append(float(loss.detach())); dzi.append(gi); dzj.append(gj)
    assert abs(sum(losses) - float(full)) < 5e-6
    close(torch.cat(dzi), zi_r.grad, tol=2e-6, what="dz_i of the 8 slices")
    close(torch.cat(dzj), zj_r.grad, tol=2e-6, what="dz_j of the 8 slices")
    # one launch over the whole global batch (what a single-GPU step at B = 2048 would run) agrees as well
    loss, gi, gj = ops.ntxent_fwd_bwd(zi_d, zj_d, tau)
    assert abs(float(loss.detach()) - float(full)) < 5e-6
    close(gi, zi_r.grad, tol=2e-6, what="dz_i global")


def test_batched_index_select_symbol(ops):
    """encoder/gcn_lib/torch_nn.py:79-98 restated with torch ops: flat gather of (B, C, N, 1) by (B, N, k) -> (B, C, N, k)"""
    from neuralsampleid_amd.encoder.gcn_lib.torch_nn import batched_index_select
    B, C, N, k = 3, 20, 48, 5
    x = rnd("bisx", B, C, N, 1)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:k].repeat(N, 1) for b in range(B)])
    idx = (idx + torch.arange(N).view(1, N, 1)) % N
    xr = x.clone().requires_grad_(True)
    idx_base = torch.arange(B).view(-1, 1, 1) * N
    flat = xr.transpose(2, 1).reshape(B * N, C)[(idx + idx_base).view(-1)]
    ref = flat.view(B, N, k, C).permute(0, 3, 1, 2).contiguous()
    gout = rnd("bisg", B, C, N, k)
    ref.backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    got = batched_index_select(xd, idx.to(DEV))
    got.backward(gout.to(DEV))
    assert got.shape == (B, C, N, k) and torch.equal(got.detach().cpu(), ref.detach())
    close(xd.grad.reshape(B, C, N, 1), xr.grad, tol=1e-6, what="batched_index_select dx")
    z = ops.zeros((5, 7), DEV)
    assert torch.equal(z.cpu(), torch.zeros(5, 7))
    buf = torch.full((1031,), 3.0, device=DEV)
    ops.fill_zero(buf[:1030])                        # 257 16-byte granules + an 8-byte tail; the bytes beyond stay
    assert float(buf[:1030].abs().sum()) == 0 and float(buf[1030:].sum()) == 3.0
    s2 = torch.tensor([0.5], device=DEV)
    assert torch.equal(ops.scale_f32(torch.arange(9.0, device=DEV), s2).cpu(), torch.arange(9.0) * 0.5)


def test_clip_adam(ops):
    n = 100003
    P = {"w": rnd("adp", n)}
    opt = R.AdamState(P, lr=8e-5)
    p = P["w"].clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([8e-5, 0.9, 0.999, 1e-8, 1.0], device=DEV)
    step = torch.zeros((), dtype=torch.int64, device=DEV)
    gn = torch.zeros(1, device=DEV)
    for it in range(3):
        g = rnd(f"adg{it}", n) * (3.0 if it < 2 else 1e-3)      # clipped twice, then unclipped
        total = R.clip_and_adam(P, {"w": g}, opt, 1.0)
        gd = g.to(DEV)
        ops.adam_step(p, gd, m, v, hyper, step, ops.sumsq_partial(gd), gn)
        assert abs(float(gn) - total) / total < 1e-5
        close(p, P["w"], tol=1e-6, what=f"adam step {it}")
    assert int(step) == 3


def test_layout_roundtrip(ops):
    x = rnd("lay", 3, 70, 50)
    rows = ops.bcn_to_rows(x.to(DEV))
    assert torch.equal(rows.cpu().reshape(3, 50, 70), x.transpose(1, 2))
    assert torch.equal(ops.rows_to_bcn(rows, 3, 50).cpu(), x)
