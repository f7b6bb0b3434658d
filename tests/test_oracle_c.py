"""Pins the plain-C oracle (oracle/c/nsid_oracle.c: every loop spelled out, no torch kernels) to the golden vectors produced by the
reference's own modules, and to the torch oracle (oracle/ref_torch.py) on seeded inputs — the "CPU restatement behind the C ABI's
conventions" of SURVEY.md section 8b. Test infrastructure: the product path never loads this library."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from compare import allclose
from conftest import from_rows, to_rows
from oracle import ref_torch as R
from synth import GRAFP_CFG, synth_tensor
from test_oracle_golden import KNN_CASES, knn_set_mismatch, synth_P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.set_num_threads(8)

FP = ctypes.POINTER(ctypes.c_float)
IP = ctypes.POINTER(ctypes.c_int32)
UP = ctypes.POINTER(ctypes.c_uint8)


@pytest.fixture(scope="module")
def C():
    so = os.path.join(ROOT, "oracle", "_build", "libnsid_oracle.so")
    src = os.path.join(ROOT, "oracle", "c", "nsid_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "c")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.oracle_knn_graph.argtypes = [FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, IP]
    lib.oracle_mr_aggregate_fwd.argtypes = [FP, IP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP, UP]
    lib.oracle_mr_aggregate_bwd.argtypes = [FP, IP, UP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP]
    lib.oracle_linear_fwd.argtypes = [FP, FP, FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP]
    lib.oracle_bn_fwd.argtypes = [FP, ctypes.c_int, ctypes.c_int, FP, FP, FP, FP, ctypes.c_int, FP]
    lib.oracle_downsample3_fwd.argtypes = [FP, FP, FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP]
    lib.oracle_peak_patchify_fwd.argtypes = [FP, FP, FP] + [ctypes.c_int] * 6 + [FP]
    lib.oracle_ntxent.argtypes = [FP, FP, ctypes.c_int, ctypes.c_int, ctypes.c_float, FP, FP, FP]
    assert lib.oracle_version() == 1
    return lib


def f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)


def p(a, ty=FP):
    return a.ctypes.data_as(ty) if a is not None else None


@pytest.mark.parametrize("tag,kds", KNN_CASES)
def test_knn_graph(C, golden, tag, kds):
    """K1 against the reference's dense_knn_matrix goldens: neighbour sets identical outside the recorded near-ties, self first"""
    g = golden("knn_" + tag)
    y = f32(to_rows(g.t("x")))
    B, N, Cc = y.shape
    for k, d in kds:
        idx = np.zeros((B, N, k), np.int32)
        assert C.oracle_knn_graph(p(y), B, N, Cc, k, d, p(idx, IP)) == 0
        gap = g[f"mingap_k{k}_d{d}"] if d > 1 else g[f"setgap_k{k}_d{d}"]
        hard, soft = knn_set_mismatch(idx, g[f"idx_k{k}_d{d}"], gap)
        assert hard == 0, (k, d, hard, soft)
        assert (idx[..., 0] == np.arange(N)).all()
    assert C.oracle_knn_graph(p(y), B, N, Cc, N, 2, p(np.zeros((B, N, N), np.int32), IP)) == -1      # k * dilation > N


def test_mr_aggregate(C, golden):
    """K2 forward exact (gather, subtract, max), backward against the reference's autograd"""
    g = golden("mragg_c64n256")
    y = f32(to_rows(g.t("x")))
    B, N, Cc = y.shape
    idx = np.ascontiguousarray(g["idx"], dtype=np.int32)
    k = idx.shape[-1]
    u = np.zeros((B * N, 2 * Cc), np.float32)
    am = np.zeros((B * N, Cc), np.uint8)
    assert C.oracle_mr_aggregate_fwd(p(y), p(idx, IP), B, N, Cc, k, p(u), p(am, UP)) == 0
    assert torch.equal(from_rows(torch.from_numpy(u).reshape(B, N, 2 * Cc)), g.t("u"))
    gu = f32(to_rows(g.t("gu")))
    dy = np.zeros((B * N, Cc), np.float32)
    assert C.oracle_mr_aggregate_bwd(p(gu), p(idx, IP), p(am, UP), B, N, Cc, k, p(dy)) == 0
    assert allclose(from_rows(torch.from_numpy(dy).reshape(B, N, Cc)), g.t("dx"), atol=1e-6)


def test_downsample_and_batchnorm(C, golden):
    """K5 + BatchNorm: eval output, train output and the updated running statistics of the reference's Downsample"""
    g = golden("downsample_c64n256")
    Cc = 64
    P = synth_P({"conv.0.weight": (2 * Cc, Cc, 3, 3), "conv.0.bias": (2 * Cc,), "conv.1.weight": (2 * Cc,), "conv.1.bias": (2 * Cc,),
                 "conv.1.running_mean": (2 * Cc,), "conv.1.running_var": (2 * Cc,)}, "ds.")
    x = f32(to_rows(g.t("x")))
    B, N, _ = x.shape
    No = (N - 1) // 2 + 1
    conv = np.zeros((B * No, 2 * Cc), np.float32)
    assert C.oracle_downsample3_fwd(p(x), p(f32(P["conv.0.weight"])), p(f32(P["conv.0.bias"])), B, N, Cc, 2 * Cc, p(conv)) == 0
    for training, key in ((0, "y_eval"), (1, "y_train")):
        rm, rv = f32(P["conv.1.running_mean"]).copy(), f32(P["conv.1.running_var"]).copy()
        out = np.zeros_like(conv)
        assert C.oracle_bn_fwd(p(conv), B * No, 2 * Cc, p(f32(P["conv.1.weight"])), p(f32(P["conv.1.bias"])), p(rm), p(rv), training,
                               p(out)) == 0
        assert allclose(from_rows(torch.from_numpy(out).reshape(B, No, 2 * Cc)), g.t(key), atol=2e-5)
        if training:
            assert torch.allclose(torch.from_numpy(rm), g.t("post.conv.1.running_mean"), atol=1e-6)
            assert torch.allclose(torch.from_numpy(rv), g.t("post.conv.1.running_var"), atol=1e-6)


def test_peak_patchify(C, golden):
    g = golden("peak_b8")
    w = f32(synth_tensor("peak_extractor.convs.0.weight", torch.empty(8, 3, 4, 8)))
    b = f32(synth_tensor("peak_extractor.convs.0.bias", torch.empty(8)))
    x = f32(g.t("x"))
    B, H, W = x.shape
    out = np.zeros((B * 256, 8), np.float32)
    assert C.oracle_peak_patchify_fwd(p(x), p(w), p(b), B, H, W, 4, 8, 8, p(out)) == 0
    assert allclose(torch.from_numpy(out).reshape(B, 256, 8).transpose(1, 2), g.t("y"), atol=1e-5)


@pytest.mark.parametrize("B", [2, 8, 256])
def test_ntxent(C, golden, B):
    """K6 forward and dz against the reference's 2B-iteration loop + autograd"""
    g = golden(f"ntxent_b{B}")
    zi, zj = f32(g.t("z_i")), f32(g.t("z_j"))
    loss = np.zeros(1, np.float32)
    dzi, dzj = np.zeros_like(zi), np.zeros_like(zj)
    assert C.oracle_ntxent(p(zi), p(zj), B, zi.shape[1], float(g["tau"]), p(loss), p(dzi), p(dzj)) == 0
    assert abs(float(loss[0]) - float(g["loss"][0])) < 2e-6
    assert allclose(torch.from_numpy(dzi), g.t("dz_i"), atol=1e-6)
    assert allclose(torch.from_numpy(dzj), g.t("dz_j"), atol=1e-6)


def test_grouped_linear_and_block_against_the_torch_oracle(C):
    """K3 / K4 on seeded data: the grouped 1x1 conv, and a whole eval-mode Grapher + FFN block (fc1 -> kNN -> max-relative -> grouped
    conv + BN + ReLU -> fc2 + shortcut -> FFN) assembled from the C functions, against oracle/ref_torch.py with the same neighbour
    ids (its own search must agree with the C search on every row whose distances are separated)"""
    Cc, N, B, k = 64, 256, 2, 3
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, N, Cc, generator=gen)
    shapes = {}
    def bn(pre, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[pre + n] = (c,)
    shapes["0.fc1.0.weight"] = (Cc, Cc, 1, 1); shapes["0.fc1.0.bias"] = (Cc,); bn("0.fc1.1.", Cc)
    shapes["0.graph_conv.gconv.nn.0.weight"] = (2 * Cc, Cc // 2, 1, 1); shapes["0.graph_conv.gconv.nn.0.bias"] = (2 * Cc,)
    bn("0.graph_conv.gconv.nn.1.", 2 * Cc)
    shapes["0.fc2.0.weight"] = (Cc, 2 * Cc, 1, 1); shapes["0.fc2.0.bias"] = (Cc,); bn("0.fc2.1.", Cc)
    shapes["1.fc1.0.weight"] = (4 * Cc, Cc, 1, 1); bn("1.fc1.1.", 4 * Cc)
    shapes["1.fc2.0.weight"] = (Cc, 4 * Cc, 1, 1); bn("1.fc2.1.", Cc)
    P = synth_P(shapes, "blk.")
    M = B * N

    def lin(rows, pre, nout, kin, groups=1):
        out = np.zeros((rows.shape[0], groups * nout), np.float32)
        bias = f32(P[pre + "bias"]) if pre + "bias" in P else None
        assert C.oracle_linear_fwd(p(rows), p(f32(P[pre + "weight"])), p(bias), rows.shape[0], nout, kin, groups, p(out)) == 0
        return out

    def bne(rows, pre):
        out = np.zeros_like(rows)
        assert C.oracle_bn_fwd(p(rows), rows.shape[0], rows.shape[1], p(f32(P[pre + "weight"])), p(f32(P[pre + "bias"])),
                               p(f32(P[pre + "running_mean"]).copy()), p(f32(P[pre + "running_var"]).copy()), 0, p(out)) == 0
        return out

    rows = f32(x.reshape(M, Cc))
    y = bne(lin(rows, "0.fc1.0.", Cc, Cc), "0.fc1.1.")
    idx = np.zeros((B, N, k), np.int32)
    assert C.oracle_knn_graph(p(y), B, N, Cc, k, 1, p(idx, IP)) == 0
    y_t = R.batchnorm_rows(R.linear_rows(x.reshape(M, Cc), P, "0.fc1.0."), P, "0.fc1.1.", False, None).reshape(B, N, Cc)
    assert np.abs(y - f32(y_t.reshape(M, Cc))).max() < 2e-5
    idx_t = R._knn_graph(y_t, k, 1)
    same = (np.sort(idx, -1) == np.sort(idx_t.numpy(), -1)).all(-1)
    assert same.mean() > 0.995                                               # the rest: near-ties under the 1e-5 feature difference
    u = np.zeros((M, 2 * Cc), np.float32)
    assert C.oracle_mr_aggregate_fwd(p(y), p(idx, IP), B, N, Cc, k, p(u), None) == 0
    g_out = lin(u, "0.graph_conv.gconv.nn.0.", Cc // 2, Cc // 2, groups=4)
    ref_g = R.grouped_linear(R.mr_aggregate(y_t, torch.from_numpy(idx).long()).reshape(M, 2 * Cc), P, "0.graph_conv.gconv.nn.0.")
    assert np.abs(g_out - f32(ref_g)).max() < 5e-5
    v = np.maximum(bne(g_out, "0.graph_conv.gconv.nn.1."), 0)
    x1 = bne(lin(v, "0.fc2.0.", Cc, 2 * Cc), "0.fc2.1.") + rows
    h = np.maximum(bne(lin(x1, "1.fc1.0.", 4 * Cc, Cc), "1.fc1.1."), 0)
    out = bne(lin(h, "1.fc2.0.", Cc, 4 * Cc), "1.fc2.1.") + x1
    R.TAPE = R.KnnTape(replay=[torch.from_numpy(idx)])
    try:
        ref = R.ffn(R.grapher(x, P, "0.", k, 1, False, None), P, "1.", False, None)
    finally:
        R.TAPE = None
    err = np.abs(out - f32(ref.reshape(M, Cc))).max() / max(1.0, float(ref.abs().max()))
    assert err < 5e-5, err
