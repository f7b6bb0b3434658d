"""The C ABI of libnsid_hip.so beside the plain-C oracle (oracle/c/nsid_oracle.c), through RAW POINTERS: both shared libraries are loaded
with ctypes here, argument lists are written out from include/nsid.h, and nothing of neuralsampleid_amd/ops.py or _lib.py is on the call
path (torch only owns the device buffers) -- what a maintainer's own binding of the header would do (VERDICT r4 task 4).
Inputs are the reference's golden inputs (tests/golden, made by the imported reference), arithmetic fp32 / exact-fp32 MFMA:

  nsid_knn_graph          vs oracle_knn_graph          torch_edge.py:270-284, 70-103    neighbour SETS outside recorded near-ties
  nsid_mr_aggregate_fwd   vs oracle_mr_aggregate_fwd   torch_vertex.py:21-32            bit-exact (gather, subtract, max), arg-max bytes
  nsid_mr_aggregate_bwd   vs oracle_mr_aggregate_bwd   its autograd                     1e-6
  nsid_linear_fwd (g = 4) vs oracle_linear_fwd         torch_nn.py:52-76 grouped conv   5e-6 relative
  nsid_downsample3_fwd    vs oracle_downsample3_fwd    graph_encoder.py:44              5e-6 relative
  nsid_peak_patchify_fwd  vs oracle_peak_patchify_fwd  peak_extractor.py:45-70          1e-5
  nsid_ntxent_fwd_bwd     vs oracle_ntxent             simclr/ntxent.py:5-30            2e-6 / 1e-6"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import to_rows
from synth import synth_tensor
from test_oracle_golden import KNN_CASES, knn_set_mismatch, synth_P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
I, F, P_, L = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_long
FP, IP, UP = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
NSID_F32 = 0


@pytest.fixture(scope="module")
def libs():
    """(libnsid_hip.so, libnsid_oracle.so) bound from their headers; fp32 arithmetic on the GPU side"""
    hip = ctypes.CDLL(os.path.join(ROOT, "neuralsampleid_amd", "libnsid_hip.so"))
    so = os.path.join(ROOT, "oracle", "_build", "libnsid_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "c")], stdout=subprocess.DEVNULL)
    orc = ctypes.CDLL(so)
    # include/nsid.h
    hip.nsid_knn_graph.argtypes = [P_, I, P_, P_, I, I, I, I, I, P_, I, P_]
    hip.nsid_mr_aggregate_fwd.argtypes = [P_, I, P_, P_, P_, I, I, I, I, P_, P_, I, P_]
    hip.nsid_mr_aggregate_bwd.argtypes = [P_, P_, P_, I, I, I, I, P_, I, P_]
    hip.nsid_linear_fwd.argtypes = [P_, I, P_, I, P_, P_, I, I, I, I, I, P_, P_, I, I, P_, I, I, P_]
    hip.nsid_pack_ds_weight.argtypes = [P_, I, I, P_, P_]
    hip.nsid_downsample3_fwd.argtypes = [P_, I, I, I, P_, I, P_, P_, I, P_, I, P_]
    hip.nsid_peak_patchify_fwd.argtypes = [P_, P_, P_, I, I, I, I, I, I, P_, I, P_, I, P_]
    hip.nsid_ntxent_ws_floats.argtypes = [I]
    hip.nsid_ntxent_ws_floats.restype = ctypes.c_size_t
    hip.nsid_ntxent_fwd_bwd.argtypes = [P_, P_, I, I, F, I, I, P_, P_, P_, P_, P_]
    hip.nsid_set_gemm_precision.argtypes = [I]
    hip.nsid_get_gemm_precision.restype = I
    # oracle/c/nsid_oracle.c
    orc.oracle_knn_graph.argtypes = [FP, I, I, I, I, I, IP]
    orc.oracle_mr_aggregate_fwd.argtypes = [FP, IP, I, I, I, I, FP, UP]
    orc.oracle_mr_aggregate_bwd.argtypes = [FP, IP, UP, I, I, I, I, FP]
    orc.oracle_linear_fwd.argtypes = [FP, FP, FP, I, I, I, I, FP]
    orc.oracle_downsample3_fwd.argtypes = [FP, FP, FP, I, I, I, I, FP]
    orc.oracle_peak_patchify_fwd.argtypes = [FP, FP, FP] + [I] * 6 + [FP]
    orc.oracle_ntxent.argtypes = [FP, FP, I, I, F, FP, FP, FP]
    keep = hip.nsid_get_gemm_precision()
    assert hip.nsid_set_gemm_precision(0) == 0                 # NSID_GEMM_FP32: exact-fp32 MFMA, the parity arithmetic
    yield hip, orc
    hip.nsid_set_gemm_precision(keep)


def f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)


def hp(a, ty=FP):
    return a.ctypes.data_as(ty) if a is not None else None


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def dp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run(rc):
    assert rc == 0, rc
    torch.cuda.synchronize()


@pytest.mark.parametrize("tag,kds", KNN_CASES)
def test_knn_graph_both_libraries_on_the_reference_inputs(libs, golden, tag, kds):
    hip, orc = libs
    g = golden("knn_" + tag)
    y = f32(to_rows(g.t("x")))
    B, N, C = y.shape
    yd = dev(y)
    for k, d in kds:
        io = np.zeros((B, N, k), np.int32)
        assert orc.oracle_knn_graph(hp(y), B, N, C, k, d, hp(io, IP)) == 0
        ih = torch.zeros((B, N, k), dtype=torch.int32, device=DEV)
        run(hip.nsid_knn_graph(dp(yd), C, None, None, B, N, C, k, d, dp(ih), NSID_F32, None))
        ih = ih.cpu().numpy()
        gap = g[f"mingap_k{k}_d{d}"] if d > 1 else g[f"setgap_k{k}_d{d}"]
        for name, idx in (("hip", ih), ("oracle", io)):
            hard, soft = knn_set_mismatch(idx, g[f"idx_k{k}_d{d}"], gap)
            assert hard == 0, (name, k, d, hard, soft)
        # and against each other: identical sets on every row the reference's own margin separates
        hard, _ = knn_set_mismatch(ih, io, gap)
        assert hard == 0 and (ih[..., 0] == np.arange(N)).all()


def test_max_relative_aggregation_both_libraries(libs, golden):
    hip, orc = libs
    g = golden("mragg_c64n256")
    y = f32(to_rows(g.t("x")))
    B, N, C = y.shape
    idx = np.ascontiguousarray(g["idx"], dtype=np.int32)
    k = idx.shape[-1]
    uo, ao = np.zeros((B * N, 2 * C), np.float32), np.zeros((B * N, C), np.uint8)
    assert orc.oracle_mr_aggregate_fwd(hp(y), hp(idx, IP), B, N, C, k, hp(uo), hp(ao, UP)) == 0
    yd, id_ = dev(y), dev(idx)
    uh = torch.zeros((B * N, 2 * C), device=DEV)
    ah = torch.zeros((B * N, C), dtype=torch.uint8, device=DEV)
    run(hip.nsid_mr_aggregate_fwd(dp(yd), C, None, None, dp(id_), B, N, C, k, dp(uh), dp(ah), NSID_F32, None))
    assert np.array_equal(uh.cpu().numpy(), uo)                                       # gather, subtract, max: no rounding freedom
    assert np.array_equal(ah.cpu().numpy(), ao)                                       # first maximum on ties, as torch.max
    gu = f32(to_rows(g.t("gu")))
    dyo = np.zeros((B * N, C), np.float32)
    assert orc.oracle_mr_aggregate_bwd(hp(gu), hp(idx, IP), hp(ao, UP), B, N, C, k, hp(dyo)) == 0
    dyh = torch.zeros((B * N, C), device=DEV)
    gud = dev(gu)                    # (every device buffer keeps a name until the launch has run: a temporary would be freed, and its
    run(hip.nsid_mr_aggregate_bwd(dp(gud), dp(id_), dp(ah), B, N, C, k, dp(dyh), NSID_F32, None))      # block reused, behind a raw pointer)
    assert np.abs(dyh.cpu().numpy() - dyo).max() < 1e-6


def test_grouped_linear_both_libraries(libs, golden):
    """the grouped 1x1 conv of MRConv2d (groups = 4) on the aggregation golden's u: x (M, 4 * K) -> out (M, 4 * Nout)"""
    hip, orc = libs
    g = golden("mragg_c64n256")
    u = f32(to_rows(g.t("u")))
    B, N, C2 = u.shape
    M, G, K = B * N, 4, C2 // 4
    P = synth_P({"w": (C2, K, 1, 1), "b": (C2,)}, "cabi.gconv.")
    w, b = f32(P["w"].reshape(C2, K)), f32(P["b"])
    oo = np.zeros((M, C2), np.float32)
    assert orc.oracle_linear_fwd(hp(u.reshape(M, C2)), hp(w), hp(b), M, K, K, G, hp(oo)) == 0
    oh = torch.zeros((M, C2), device=DEV)
    ud, wd, bd = dev(u.reshape(M, C2)), dev(w), dev(b)
    run(hip.nsid_linear_fwd(dp(ud), C2, dp(wd), NSID_F32, dp(bd), dp(oh), C2, M, K, K, G, None, None, 0, 0, None, 1, NSID_F32, None))
    assert np.abs(oh.cpu().numpy() - oo).max() <= 5e-6 * max(1.0, np.abs(oo).max())


def test_downsample_both_libraries(libs, golden):
    hip, orc = libs
    g = golden("downsample_c64n256")
    C = 64
    P = synth_P({"conv.0.weight": (2 * C, C, 3, 3), "conv.0.bias": (2 * C,)}, "ds.")
    x = f32(to_rows(g.t("x")))
    B, N, _ = x.shape
    No = N // 2
    w, b = f32(P["conv.0.weight"]), f32(P["conv.0.bias"])
    oo = np.zeros((B * No, 2 * C), np.float32)
    assert orc.oracle_downsample3_fwd(hp(x), hp(w), hp(b), B, N, C, 2 * C, hp(oo)) == 0
    wd = dev(w)
    wp = torch.zeros((2 * C, 3 * C), device=DEV)
    run(hip.nsid_pack_ds_weight(dp(wd), 2 * C, C, dp(wp), None))
    oh = torch.zeros((B * No, 2 * C), device=DEV)
    xd, bd = dev(x.reshape(B * N, C)), dev(b)
    run(hip.nsid_downsample3_fwd(dp(xd), B, N, C, dp(wp), NSID_F32, dp(bd), dp(oh), 2 * C, None, NSID_F32, None))
    assert np.abs(oh.cpu().numpy() - oo).max() <= 5e-6 * max(1.0, np.abs(oo).max())


def test_peak_patchify_both_libraries(libs, golden):
    hip, orc = libs
    g = golden("peak_b8")
    w = f32(synth_tensor("peak_extractor.convs.0.weight", torch.empty(8, 3, 4, 8)))
    b = f32(synth_tensor("peak_extractor.convs.0.bias", torch.empty(8)))
    x = f32(g.t("x"))
    B, H, W = x.shape
    oo = np.zeros((B * 256, 8), np.float32)
    assert orc.oracle_peak_patchify_fwd(hp(x), hp(w), hp(b), B, H, W, 4, 8, 8, hp(oo)) == 0
    oh = torch.zeros((B * 256, 8), device=DEV)
    mm = torch.zeros((B, 2), device=DEV)
    xd, wd, bd = dev(x), dev(w), dev(b)
    run(hip.nsid_peak_patchify_fwd(dp(xd), dp(wd), dp(bd), B, H, W, 4, 8, 8, dp(oh), 8, dp(mm), NSID_F32, None))
    assert np.abs(oh.cpu().numpy() - oo).max() < 1e-5
    assert np.abs(oh.cpu().numpy().reshape(B, 256, 8).transpose(0, 2, 1) - g["y"]).max() < 1e-5      # and the reference's own output


@pytest.mark.parametrize("B", [2, 8, 256])
def test_ntxent_both_libraries(libs, golden, B):
    hip, orc = libs
    g = golden(f"ntxent_b{B}")
    zi, zj = f32(g.t("z_i")), f32(g.t("z_j"))
    d, tau = zi.shape[1], float(g["tau"])
    lo, dio, djo = np.zeros(1, np.float32), np.zeros_like(zi), np.zeros_like(zj)
    assert orc.oracle_ntxent(hp(zi), hp(zj), B, d, tau, hp(lo), hp(dio), hp(djo)) == 0
    ws = torch.zeros(int(hip.nsid_ntxent_ws_floats(B)), device=DEV)
    lh = torch.zeros(1, device=DEV)
    dih, djh = torch.zeros((B, d), device=DEV), torch.zeros((B, d), device=DEV)
    zid, zjd = dev(zi), dev(zj)
    run(hip.nsid_ntxent_fwd_bwd(dp(zid), dp(zjd), B, d, tau, 0, B, dp(ws), dp(lh), dp(dih), dp(djh), None))
    assert abs(float(lh[0]) - float(lo[0])) < 2e-6 and abs(float(lh[0]) - float(g["loss"][0])) < 2e-6
    assert np.abs(dih.cpu().numpy() - dio).max() < 1e-6 and np.abs(djh.cpu().numpy() - djo).max() < 1e-6
