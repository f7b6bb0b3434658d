"""Graphs beyond grafp.yaml's 256 nodes (a cfg with n_mels = 256 is legal for GraphEncoder — encoder/graph_encoder.py:144
takes N = n_mels * n_frames / (patch_bins * patch_frames) = 1 024, stages of 1 024 / 512 / 256 / 128 nodes, a clip of 65 536 features
at every stage). No golden exists for it (the reference was never run there); the oracle, pinned to the reference on the standard
graphs, is the checker. What runs: knn_big_kernel (a strip per workgroup, csrc/knn.hip), the grid form of the aggregation forward, the
edge-list-only aggregation backward in bf16 / the atomic scatter in fp32 (csrc/mr.hip), the peak extractor with a 139 KB clip in LDS
and its backward in bands of patch rows (csrc/misc.hip)."""
import numpy as np
import pytest
import torch

from compare import maxerr, relerr
from oracle import ref_torch as R
from synth import GRAFP_CFG, synth_randn, synth_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(GRAFP_CFG, n_mels=256)


@pytest.fixture()
def restore_mode():
    yield
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    F_.TAPE = None
    R.TAPE, R.STORAGE = None, None
    ops.set_gemm_precision("fp32")
    F_.set_activation_dtype("fp32")


def set_mode(mode):
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    ops.set_gemm_precision(mode)
    F_.set_activation_dtype(mode)


@pytest.mark.parametrize("N,C,k,d", [(1024, 64, 3, 1), (512, 128, 5, 1), (1024, 64, 9, 2), (512, 128, 18, 2), (320, 64, 4, 1),
                                     (256, 256, 3, 1), (128, 512, 18, 1), (1024, 64, 18, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_knn_beyond_256_nodes_or_32768_features(N, C, k, d, dtype):
    """ids against the oracle's (torch_edge.py:96-103 restated): self first, every row's set equal except where the k d-th and the next
    distance of that row lie within 1e-5 of each other (a near-tie: either answer is a top-k of the same distances)"""
    from neuralsampleid_amd import ops
    B = 2
    x = synth_randn(f"big{N}x{C}", B, N, C)
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    ref = R._knn_graph(x, k, d).numpy()
    ops.launch_counters(reset=True)
    idx = ops.knn_graph(x.reshape(B * N, C).to(DEV).to(dtype).contiguous(), B, N, C, k, d).cpu().long().numpy()
    assert ops.launch_counters()["knn_big"] == 1
    assert idx.shape == (B, N, k) and (idx[..., 0] == np.arange(N)).all() and idx.min() >= 0 and idx.max() < N
    bad = (np.sort(idx, -1) != np.sort(ref, -1)).any(-1)
    assert bad.mean() < 5e-3, bad.mean()
    xn = torch.nn.functional.normalize(x.double(), dim=-1)
    D = (2.0 - 2.0 * xn @ xn.transpose(1, 2)).numpy()
    for b, n in np.argwhere(bad):
        own, want = np.sort(D[b, n, idx[b, n]]), np.sort(D[b, n, ref[b, n]])
        assert np.abs(own - want).max() < 1e-5, (b, n, own, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_peak_extractor_on_256_mel_bins(dtype):
    """forward (min-max, ramps, patch conv, ReLU) and the parameter gradients (the backward walks the clip in four bands of patch rows)
    against the oracle's autograd, 1 024 patches per clip"""
    from neuralsampleid_amd import ops
    B, H, W, pb, pf, F = 3, 256, 128, 4, 8, 8
    spec = synth_randn("pk256", B, H, W) * 20 - 40
    w = (synth_randn("pk256w", F, 3, pb, pf) * 0.1).requires_grad_(True)
    bias = (synth_randn("pk256b", F) * 0.1).requires_grad_(True)
    y_ref = R.peak_patchify(spec, {"p.convs.0.weight": w, "p.convs.0.bias": bias}, "p.", dict(CFG, patch_bins=pb, patch_frames=pf))
    gout = synth_randn("pk256g", *y_ref.shape).to(dtype).float()          # the upstream gradient as the storage type holds it
    y_ref.backward(gout)
    out, minmax = ops.peak_patchify_fwd(spec.to(DEV), w.detach().to(DEV), bias.detach().to(DEV), pb, pf, dtype)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert maxerr(out.reshape(B, -1, F).float(), y_ref.detach()) < tol * max(1.0, float(y_ref.abs().max()))
    # backward on the reference's output (in the storage type: the ReLU mask is the reference's) and the same upstream gradient
    dw, db = torch.zeros(F, 3, pb, pf, device=DEV), torch.zeros(F, device=DEV)
    o_s = y_ref.detach().reshape(-1, F).to(DEV).to(dtype).contiguous()
    g_s = gout.reshape(-1, F).to(DEV).to(dtype).contiguous()
    ops.peak_patchify_bwd(spec.to(DEV), minmax, o_s, g_s, pb, pf, dw, db)
    assert relerr(dw, w.grad) < 1e-4, relerr(dw, w.grad)
    assert relerr(db, bias.grad) < 1e-4, relerr(db, bias.grad)


def _model():
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t"))
    model.load_state_dict(synth_state(model.state_dict()))
    return model


def _clips(B):
    x_i = synth_randn("bigx_i", B, 256, CFG["n_frames"]) * 20 - 40
    return x_i, x_i + 3 * synth_randn("bigx_j", B, 256, CFG["n_frames"])


def _oracle(P, x_i, x_j, training, storage=None):
    """(h_i, z_i, z_j, loss, tape, P with gradients) of the oracle, its own graphs recorded"""
    plan = R.encoder_plan("t", 3, n_nodes=1024)
    R.TAPE, R.STORAGE = R.KnnTape(), storage
    try:
        if training:
            for k_ in R.trainable_keys(P):
                P[k_].requires_grad_(True)
            h_i, _, z_i, z_j = R.simclr_forward(x_i, x_j, P, CFG, plan, True, R.BNState())
            loss = R.ntxent(z_i, z_j, CFG["tau"])
            loss.backward()
        else:
            with torch.no_grad():
                h_i, _, z_i, z_j = R.simclr_forward(x_i, x_j, P, CFG, plan, False)
                loss = R.ntxent(z_i, z_j, CFG["tau"])
        return h_i.detach(), z_i.detach(), z_j.detach(), float(loss.detach()), list(R.TAPE.recorded)
    finally:
        R.TAPE, R.STORAGE = None, None


# bounds: ~3x the values measured on MI355X (round 6, tools/big_graph_probe.py): fp32 eval max |dz| 6.0e-7, step 0 |dloss| 9.5e-7, global
# gradient norm 1.7e-3 (B = 2 through 60 train-mode BatchNorms: the floor the B = 4 / B = 8 tests see too), last-layer gradient 3.2e-5
TOL32 = {"eval_dz": 5e-6, "eval_h": 5e-6, "dloss": 5e-6, "dz": 1e-5, "gnorm": 6e-3, "late": 1e-4}


def test_model_on_1024_node_graphs_fp32(restore_mode):
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    set_mode("fp32")
    model = _model()
    P = {n: v.clone() for n, v in model.state_dict().items() if "relative_pos" not in n}
    model = model.to(DEV)
    x_i, x_j = _clips(2)
    h_r, z_r, _, _, tape = _oracle(P, x_i, x_j, False)
    assert [tuple(t.shape)[1] for t in tape[:12]] == [1024] * 2 + [512] * 2 + [256] * 6 + [128] * 2
    model.eval()
    ops.launch_counters(reset=True)
    F_.TAPE = F_.KnnTape(replay=tape)
    with torch.no_grad():
        h, _, z, _ = model(x_i.to(DEV), x_j.to(DEV))
    own = F_.TAPE.recorded
    F_.TAPE = None
    cnt = ops.launch_counters()
    assert cnt["knn_big"] == 24 and cnt["knn2"] == 0, cnt                     # every stage holds 65 536 features per clip
    same = [float((np.sort(a.cpu().numpy(), -1) == np.sort(b.numpy(), -1)).all(-1).mean()) for a, b in zip(own, tape)]
    assert min(same) > 0.99, same                                             # own search = the oracle's, up to near-ties
    assert maxerr(z, z_r) < TOL32["eval_dz"] and relerr(h, h_r) < TOL32["eval_h"]

    h_r, z_r, _, loss_r, tape = _oracle(P, x_i, x_j, True)
    model.train()
    F_.TAPE = F_.KnnTape(replay=tape)
    model.zero_grad()
    _, _, z_i, z_j = model(x_i.to(DEV), x_j.to(DEV))
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    F_.TAPE = None
    keys = [k_ for k_ in R.trainable_keys(P) if P[k_].grad is not None]
    gn_r = float(torch.sqrt(sum(P[k_].grad.double().pow(2).sum() for k_ in keys)))
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    gn = float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values())))
    m = {"dloss": abs(float(loss.detach()) - loss_r), "dz": maxerr(z_i, z_r), "gnorm": abs(gn - gn_r) / gn_r,
         "late": max(relerr(grads[n], P[n].grad) for n in ("projector.2.weight", "projector.0.weight", "encoder.proj.weight")),
         "peak": relerr(grads["peak_extractor.convs.0.weight"], P["peak_extractor.convs.0.weight"].grad)}
    print("measured", m)
    assert m["dloss"] < TOL32["dloss"] and m["dz"] < TOL32["dz"] and m["gnorm"] < TOL32["gnorm"] and m["late"] < TOL32["late"], m
    assert m["peak"] < 0.1, m                      # the earliest layer of all: the reference's own fp32 floor at B = 2


def test_model_on_1024_node_graphs_bf16_storage(restore_mode):
    """the timed arithmetic on the large graphs, against the oracle evaluated with the same rounding points (STORAGE = "bf16"), the
    oracle's graphs forced; the aggregation backward of a 65 536-feature clip in bf16 is the edge-list-only form"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    model = _model()
    P = {n: v.clone() for n, v in model.state_dict().items() if "relative_pos" not in n}
    x_i, x_j = _clips(2)
    h_r, z_r, _, loss_r, tape = _oracle(P, x_i, x_j, True, storage="bf16")
    set_mode("bf16")
    model = model.to(DEV).train()
    F_.TAPE = F_.KnnTape(replay=tape)
    model.zero_grad()
    h_i, _, z_i, z_j = model(x_i.to(DEV), x_j.to(DEV))
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    F_.TAPE = None
    keys = [k_ for k_ in R.trainable_keys(P) if P[k_].grad is not None]
    gn_r = float(torch.sqrt(sum(P[k_].grad.double().pow(2).sum() for k_ in keys)))
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    gn = float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values())))
    cos = float(torch.nn.functional.cosine_similarity(z_i.detach().cpu().float(), z_r, dim=1).min())
    m = {"dloss": abs(float(loss.detach()) - loss_r), "cos_z_min": cos, "rel_h": relerr(h_i.float(), h_r), "gnorm": abs(gn - gn_r) / gn_r,
         "late": relerr(grads["projector.2.weight"], P["projector.2.weight"].grad)}
    print("measured", m)
    assert all(torch.isfinite(g).all() for g in grads.values())
    # measured on MI355X: |dloss| 0.035, min cos z 0.9993, h 4.0 %, global gradient norm 13 %, last-layer gradient 16 % (B = 2 under 60 train-mode
    # BatchNorms in bf16: the emulation itself sits this far from a second evaluation order); bounds = 3x
    assert m["dloss"] < 0.1 and m["cos_z_min"] > 0.995 and m["rel_h"] < 0.12 and m["gnorm"] < 0.4 and m["late"] < 0.5, m
