"""GPU parity of the module shells (reference layout in/out) and of the end-to-end SimCLR step against the goldens
generated from the reference's own modules.

kNN is discontinuous, so e2e parity is stated in two halves (see oracle.ref_torch.KnnTape):
  (a) the HIP kNN's neighbour sets equal the reference's outside recorded near-ties (margin < 1e-4);
  (b) with the neighbour indices forced to the reference's, activations/loss/gradients agree to fp32 tolerance.
Gradient tolerance follows the fp32 noise floor measured on the reference itself (tests/test_oracle_golden.py)."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, from_rows, to_rows
from oracle import ref_torch as R
import tapes
from compare import absmax, maxerr, relerr
from synth import GRAFP_CFG, synth_clips, synth_state, synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_tape():
    yield
    from neuralsampleid_amd import functional as F_
    F_.TAPE = None
    R.TAPE = None


def load_synth(module, prefix=""):
    module.load_state_dict(synth_state(module.state_dict(), prefix))
    return module.to(DEV)


def knn_mismatch(idx, gold_idx, gap, tol=1e-4):
    a = np.sort(idx.cpu().numpy(), axis=-1)
    b = np.sort(np.asarray(gold_idx), axis=-1)
    bad = (a != b).any(axis=-1)
    return int((bad & (np.asarray(gap) >= tol)).sum()), int(bad.sum())


def test_knn_module_layout(golden):
    from neuralsampleid_amd.encoder.gcn_lib.torch_edge import DenseDilatedKnnGraph
    g = golden("knn_c64n256")
    x = g.t("x").to(DEV)
    ei = DenseDilatedKnnGraph(4, 2)(x)
    assert ei.shape == (2, 2, 256, 4) and ei.dtype == torch.int64
    assert (ei[1].cpu().numpy() == g["center_k4_d2"]).all()
    hard, soft = knn_mismatch(ei[0], g["idx_k4_d2"], g["mingap_k4_d2"])
    assert hard == 0


def test_mrconv_module(golden):
    from neuralsampleid_amd.encoder.gcn_lib.torch_vertex import MRConv2d
    g = golden("mrconv_c64n256")
    m = load_synth(MRConv2d(64, 128, "relu", "batch", True), "mr.").train()
    x = g.t("x").to(DEV).requires_grad_(True)
    ei = torch.stack((g.t("idx").long(), torch.zeros_like(g.t("idx")).long())).to(DEV)
    y = m(x, ei)
    y.backward(g.t("gout").to(DEV))
    assert maxerr(y, g.t("y")) < 5e-5
    assert maxerr(x.grad, g.t("dx")) < 5e-5
    assert relerr(m.nn[0].weight.grad, g.t("dweight")) < 1e-4
    assert relerr(m.nn[1].weight.grad, g.t("dgamma")) < 1e-4
    assert relerr(m.nn[1].bias.grad, g.t("dbeta")) < 1e-4
    assert maxerr(m.nn[1].running_mean, g.t("post.nn.1.running_mean")) < 1e-6
    assert maxerr(m.nn[1].running_var, g.t("post.nn.1.running_var")) < 1e-6
    assert int(m.nn[1].num_batches_tracked) == 1


def test_dygraphconv2d_standalone_entry(golden):
    """DyGraphConv2d.forward(x (B, C, H, W)) — the reference's stand-alone entry (torch_vertex.py:126-139: reshape to
    (B, C, HW, 1), dilated kNN graph on the L2-normalised features, MRConv2d, reshape back) — against the oracle's
    composition, in eval and in training mode (forward, input gradient, running statistics)"""
    from neuralsampleid_amd.encoder.gcn_lib.torch_vertex import DyGraphConv2d
    B, C, H, W, k, d = 2, 64, 16, 16, 4, 2
    m = load_synth(DyGraphConv2d(C, 2 * C, k, d, "mr", "relu", "batch", True, False, 0.0, 1), "dy.")
    x = synth_tensor("dy_x", torch.empty(B, C, H, W))
    P = {k_[len("gconv."):]: v.detach().cpu().clone() for k_, v in m.state_dict().items()}

    def oracle(xin, training, st):
        rows = xin.reshape(B, C, H * W).transpose(1, 2)                                  # (B, N, C)
        idx = R.knn_graph(rows.detach(), k, d)
        u = R.mr_aggregate(rows, idx).reshape(B * H * W, 2 * C)
        v = torch.relu(R.batchnorm_rows(R.grouped_linear(u, P, "nn.0."), P, "nn.1.", training, st))
        return v.reshape(B, H * W, 2 * C).transpose(1, 2).reshape(B, 2 * C, H, W), idx

    m.eval()
    with torch.no_grad():
        y = m(x.to(DEV))
        y_ref, _ = oracle(x, False, None)
    assert y.shape == (B, 2 * C, H, W)
    flips = (y.cpu() - y_ref).abs().amax(dim=1) > 1e-4 * max(1.0, float(y_ref.abs().max()))   # nodes whose kNN set differs
    assert float(flips.float().mean()) < 0.01                                                     # near-ties only
    m.train()
    xg = x.clone().to(DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    st = R.BNState()
    y_ref, _ = oracle(xr, True, st)
    y = m(xg)
    gout = synth_tensor("dy_g", torch.empty_like(y_ref))
    y.backward(gout.to(DEV))
    y_ref.backward(gout)
    assert relerr(y, y_ref) < 5e-3 and relerr(xg.grad, xr.grad) < 2e-2                            # up to near-tie flips
    assert maxerr(m.gconv.nn[1].running_mean, st.updates["nn.1.running_mean"]) < 1e-3
    assert int(m.gconv.nn[1].num_batches_tracked) == 1


def test_downsample_module(golden):
    from neuralsampleid_amd.encoder.graph_encoder import Downsample
    g = golden("downsample_c64n256")
    ds = load_synth(Downsample(64, 128), "ds.")
    x = g.t("x").to(DEV)
    ds.eval()
    with torch.no_grad():
        assert maxerr(ds(x), g.t("y_eval")) < 5e-5
    ds.train()
    xg = x.clone().requires_grad_(True)
    y = ds(xg)
    y.backward(g.t("gout").to(DEV))
    assert maxerr(y, g.t("y_train")) < 5e-5
    assert maxerr(xg.grad, g.t("dx")) < 5e-5
    assert relerr(ds.conv[0].weight.grad, g.t("dweight")) < 1e-4
    assert relerr(ds.conv[1].weight.grad, g.t("dgamma")) < 1e-4
    assert maxerr(ds.conv[1].running_var, g.t("post.conv.1.running_var")) < 1e-6


BLOCKS = [("c64n256_k3d1", 64, 256, 3, 1), ("c64n256_k4d2", 64, 256, 4, 2), ("c128n128_k5d1", 128, 128, 5, 1),
          ("c512n32_k3d1", 512, 32, 3, 1), ("c64n256_k18d3", 64, 256, 18, 3),
          ("c256n64_k18d3", 256, 64, 18, 3)]          # the stage that holds 12 of config 4's 24 blocks


@pytest.mark.parametrize("tag,C,N,k,d", BLOCKS)
def test_block_modules(golden, tag, C, N, k, d):
    """Seq(Grapher, FFN) through the reference-layout forward: eval, train forward, backward, BN running stats"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.gcn_lib.torch_vertex import Grapher
    from neuralsampleid_amd.encoder.graph_encoder import FFN
    g = golden("block_" + tag)
    blk = nn.Sequential(Grapher(C, k, d, "mr", "relu", "batch", True, False, 0.2, 1, n=N, drop_path=0.0,
                                relative_pos=True), FFN(C, 4 * C, C, act="relu", drop_path=0.0))
    sd = synth_state(blk.state_dict(), "blk.")
    blk.load_state_dict(sd)
    blk.to(DEV)
    x = g.t("x")
    # the oracle (pinned to the reference on this fixture) supplies the neighbour indices for teacher forcing
    P = {k_: v.clone() for k_, v in sd.items()}
    R.TAPE = R.KnnTape()
    R.ffn(R.grapher(to_rows(x), P, "0.", k, d, False, None), P, "1.", False, None)
    idx_eval = R.TAPE.recorded[0]
    R.TAPE = R.KnnTape()
    R.ffn(R.grapher(to_rows(x), P, "0.", k, d, True, R.BNState()), P, "1.", True, R.BNState())
    idx_train = R.TAPE.recorded[0]
    R.TAPE = None

    blk.eval()
    F_.TAPE = F_.KnnTape(replay=[idx_eval])
    with torch.no_grad():
        y_eval = blk(x.to(DEV))
    own = F_.TAPE.recorded[0]
    F_.TAPE = None
    assert maxerr(y_eval, g.t("y_eval")) < 1e-4 * max(1.0, absmax(g.t("y_eval")))
    same = (np.sort(own.cpu().numpy(), -1) == np.sort(idx_eval.numpy(), -1)).all(-1)
    assert same.mean() > 0.99                                  # own kNN agrees with the oracle's up to near-ties

    blk.train()
    F_.TAPE = F_.KnnTape(replay=[idx_train])
    xg = x.to(DEV).requires_grad_(True)
    y = blk(xg)
    y.backward(g.t("gout").to(DEV))
    F_.TAPE = None
    assert maxerr(y, g.t("y_train")) < 1e-4 * max(1.0, absmax(g.t("y_train")))
    assert relerr(xg.grad, g.t("dx")) < 1e-3
    for name, p in blk.named_parameters():
        if "grad." + name in g:
            ref = g.t("grad." + name)
            if absmax(ref) < 1e-3:
                # analytically zero (a bias or BN beta whose shift the next BatchNorm removes): roundoff on both sides
                assert float(p.grad.abs().max()) < 1e-3, name
            else:
                assert relerr(p.grad, ref) < 2e-3, (name, relerr(p.grad, ref))
        elif "gradchk." + name in g:
            assert relerr(p.grad.flatten()[::997], g.t("gradsample." + name)) < 2e-3, name
    for name, b in blk.named_buffers():
        assert maxerr(b.double(), g.t("post." + name).double()) < 1e-5, name


def tape_of(g, tag):
    n = len([k_ for k_ in g if k_.startswith(f"knn.{tag}.")])
    return [g.t(f"knn.{tag}.{c}") for c in range(n)], [g[f"gap.{tag}.{c}"] for c in range(n)]


def check_tape(recorded, gold_idx, gaps):
    hard = soft = rows = 0
    for c, (r, gi, gp) in enumerate(zip(recorded, gold_idx, gaps)):
        h, s_ = knn_mismatch(r, gi, gp)
        hard, soft, rows = hard + h, soft + s_, rows + gp.size
        if h:                                                    # say which call / row, for the assertion's captured output
            a, b = np.sort(r.cpu().numpy(), axis=-1), np.sort(np.asarray(gi), axis=-1)
            for pos in np.argwhere((a != b).any(-1) & (np.asarray(gp) >= 1e-4))[:4]:
                pos = tuple(pos)
                print(f"kNN call {c} {a.shape} row {pos} gap {float(np.asarray(gp)[pos]):.3g}\n  got  {r.cpu().numpy()[pos]}\n  gold {np.asarray(gi)[pos]}")
    return hard, soft, rows


def build_model(k):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    m = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=k, size="t"))
    return load_synth(m)


@pytest.mark.parametrize("k", [3, 5])
def test_simclr_e2e(golden, k):
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    g = golden(f"e2e_b8_k{k}")
    with open(os.path.join(GOLDEN, f"e2e_b8_k{k}_checksums.json")) as f:
        chk = json.load(f)
    model = build_model(k)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)

    # ---- eval (generate.py / test_fp.py inference semantics: running statistics)
    gold_idx, gaps = tape_of(g, "eval")
    model.eval()
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    rec = F_.TAPE.recorded
    F_.TAPE = None
    assert len(rec) == 24
    hard, soft, rows = check_tape(rec, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    for got, name in ((h_i, "h_i_eval"), (h_j, "h_j_eval")):
        assert maxerr(got, g.t(name)) < 1e-4 * max(1.0, float(g.t(name).abs().max())), name
    assert maxerr(z_i, g.t("z_i_eval")) < 1e-5 and maxerr(z_j, g.t("z_j_eval")) < 1e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5

    # ---- three training steps exactly as train.py:53-75 writes them (zero_grad / forward / loss / backward /
    #      clip_grad_norm_ / Adam.step) with stock torch.optim.Adam driving our modules
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=GRAFP_CFG["lr"])
    for step in range(3):
        gold_idx, gaps = tape_of(g, f"s{step}")
        F_.TAPE = F_.KnnTape(replay=gold_idx)
        opt.zero_grad()
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        rec = F_.TAPE.recorded
        F_.TAPE = None
        if step == 0:
            hard, soft, rows = check_tape(rec, gold_idx, gaps)
            assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
            assert maxerr(h_i, g.t("h_i_train")) < 2e-4 and maxerr(h_j, g.t("h_j_train")) < 2e-4
            assert maxerr(z_i, g.t("z_i_train")) < 2e-5
            grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
            for name in [n for n in g if n.startswith("grad.")]:
                ref = g.t(name)
                if float(ref.norm()) < 1e-5:
                    assert float(grads[name[5:]].norm()) < 1e-5, name
                    continue
                late = name.startswith(("grad.encoder.backbone.14", "grad.encoder.proj", "grad.projector"))
                assert relerr(grads[name[5:]], ref) < (5e-4 if late else 4e-2), (name, relerr(grads[name[5:]], ref))
            worst = max(abs(float(grads[n].double().norm()) - nrm) / nrm
                        for n, (s_, nrm) in chk["grad"].items() if nrm > 1e-3)
            assert worst < 2e-2, worst
            sd = model.state_dict()
            for name, (s_, nrm) in chk["bn_after_step1"].items():
                assert abs(float(sd[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        print(f"e2e k={k} step {step}: |dloss| {abs(float(loss.detach()) - g['losses'][step]):.3e} "
              f"gnorm rel {abs(float(gn) - g['gnorms'][step]) / g['gnorms'][step]:.3e}")
        # measured on MI355X (round 2, k = 3 / 5): step 0 |dloss| 7e-6, gnorm 1.2e-3; steps 1-2 (after Adam updates whose
        # lr*sign(g) form turns fp32 summation-order noise on near-zero gradients into +-lr weight differences):
        # |dloss| <= 1.1e-3, gnorm <= 1.4e-2. Bounds = 3x.
        assert abs(float(loss.detach()) - g["losses"][step]) < (5e-5 if step == 0 else 3.5e-3), (step, float(loss.detach()))
        assert abs(float(gn) - g["gnorms"][step]) / g["gnorms"][step] < (5e-3 if step == 0 else 4.5e-2), (step, float(gn))


def test_deep_config4_e2e(golden):
    """BASELINE config 4 — GraphEncoder(blocks=[4,4,12,4], k=18, use_dilation=True): dilation 1/2/3/1 by stage, i.e. kNN
    through knn_sel_kernel (N = 256 / 128, k*d = 18 / 36) and knn_rank_kernel (N = 64 / 32, k*d = 54 / 18) — against the reference's own
    classes assembled with that schedule (make_golden.py::deep_reference_encoder): eval forward, then step 0 of training."""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden("deep_b4_k18")
    with open(os.path.join(GOLDEN, "deep_b4_k18_checksums.json")) as f:
        chk = json.load(f)
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=18, size="t",
                                           blocks=[4, 4, 12, 4], use_dilation=True))
    assert list(model.state_dict().keys()) == chk["keys"]                 # same module tree as the reference-built encoder
    ds = [(m.graph_conv.k, m.graph_conv.d) for m in model.modules() if type(m).__name__ == "Grapher"]
    assert ds == [(18, 1)] * 4 + [(18, 2)] * 4 + [(18, 3)] * 12 + [(18, 1)] * 4
    load_synth(model)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    # the reference's graphs: own search + the fixture's near-tie rows (30 % of the rows at k d = 18 ... 54 of 32 ... 256 nodes), every
    # clip of every build proven by its hash (tests/tapes.py)
    model.eval()
    F_.TAPE = tape = F_.KnnTape(patch=tapes.patches_of(g, "eval"))
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    F_.TAPE = None
    assert len(tape.recorded) == 48
    hard, soft, rows = tapes.check_patched(tape, g, "eval")
    assert hard == 0 and soft <= rows * 5e-3, (hard, soft, rows)
    assert maxerr(h_i, g.t("h_i_eval")) < 1e-4 * max(1.0, float(g.t("h_i_eval").abs().max()))
    assert maxerr(z_i, g.t("z_i_eval")) < 2e-5 and maxerr(z_j, g.t("z_j_eval")) < 2e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5
    model.train()
    F_.TAPE = tape = F_.KnnTape(patch=tapes.patches_of(g, "s0"))
    model.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    F_.TAPE = None
    hard, soft, rows = tapes.check_patched(tape, g, "s0")
    assert hard == 0 and soft <= rows * 5e-3, (hard, soft, rows)
    assert maxerr(h_i, g.t("h_i_train")) < 5e-4 and maxerr(z_i, g.t("z_i_train")) < 5e-5
    # B = 4 through 24 train-mode blocks: the fp32 summation-order noise of 120 BatchNorm layers reaches the loss at 8.9e-5
    # (measured on MI355X; the CPU oracle, same BLAS as the reference, sits at < 5e-5)
    assert abs(float(loss.detach()) - float(g["loss_train"][0])) < 3e-4
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    for name in [n for n in g if n.startswith("grad.")]:
        ref = g.t(name)
        if float(ref.norm()) < 1e-5:
            assert float(grads[name[5:]].norm()) < 1e-5, name
            continue
        late = name.startswith(("grad.encoder.backbone.26", "grad.encoder.proj", "grad.projector"))
        print("deep grad", name, relerr(grads[name[5:]], ref))
        assert relerr(grads[name[5:]], ref) < (1e-3 if late else 8e-2), (name, relerr(grads[name[5:]], ref))
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    assert abs(float(gn) - float(g["gnorm"][0])) / float(g["gnorm"][0]) < 2e-2
    sd = model.state_dict()
    for name, (s_, nrm) in chk["bn_after_step1"].items():
        assert abs(float(sd[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name


def test_fused_optimizer_matches_torch_adam(golden):
    """FusedClipAdam (flat buffers, 2 kernels) == clip_grad_norm_ + torch.optim.Adam on the same model and batch"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    g = golden("e2e_b8_k3")
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    losses = {}
    finals = {}
    for kind in ("torch", "fused"):
        model = build_model(3).train()
        opt = torch.optim.Adam(model.parameters(), lr=8e-5) if kind == "torch" else \
            FusedClipAdam(model.parameters(), lr=8e-5, max_norm=1.0)
        ls = []
        for step in range(3):
            gold_idx, _ = tape_of(g, f"s{step}")
            F_.TAPE = F_.KnnTape(replay=gold_idx)
            opt.zero_grad()
            _, _, z_i, z_j = model(x_i, x_j)
            loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
            loss.backward()
            F_.TAPE = None
            if kind == "torch":
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            opt.step()
            ls.append(float(loss.detach()))
        losses[kind] = ls
        finals[kind] = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    assert abs(losses["torch"][0] - losses["fused"][0]) < 1e-5
    assert abs(losses["torch"][2] - losses["fused"][2]) < 2e-2
    # one Adam step moves every weight by ~lr; agreement to a fraction of that
    n = "encoder.backbone.14.1.fc2.0.weight"
    assert maxerr(finals["torch"][n], finals["fused"][n]) < 3 * 8e-5


def test_nan_batch_is_skipped():
    """a constant clip makes (x-min)/(max-min) NaN (peak_extractor.py:48); train.py:65-68 skips such a batch"""
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    model = build_model(3).train()
    opt = FusedClipAdam(model.parameters(), lr=8e-5, max_norm=1.0)
    before = opt.flat_p.clone()
    x = torch.randn(4, 64, 128, device=DEV)
    x[1] = 3.0
    opt.zero_grad()
    _, _, z_i, z_j = model(x, x + 0.1 * torch.randn_like(x))
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    opt.step()
    assert torch.isnan(loss)
    assert torch.equal(opt.flat_p, before) and int(opt.step_count) == 0


def test_graph_encoder_reference_layout_entry(golden):
    """GraphEncoder.forward keeps the reference signature: (B, in_channels, N) -> (B, emb_dims)"""
    g = golden("e2e_b8_k3")
    model = build_model(3).eval()
    x = g.t("x_i").to(DEV)
    with torch.no_grad():
        nodes = model.peak_extractor(x)                  # (B, 8, 256) reference layout
        assert nodes.shape == (8, 8, 256)
        h = model.encoder(nodes)
        h2, _, _, _ = model(x, x)
    assert h.shape == (8, 1024)
    assert maxerr(h, h2) < 1e-4 * float(h2.abs().max())


def test_extract_fingerprints_matches_eval_forward(golden):
    """generate.py path: ragged micro-batches, eval-mode BN, one view; equals the eval goldens of the reference"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.fingerprint import extract_fingerprints, shard_bounds
    g = golden("e2e_b8_k3")
    model = build_model(3).train()                       # extract_fingerprints switches to eval and back
    x = torch.cat([g.t("x_i"), g.t("x_j")]).to(DEV)      # 16 clips
    z = extract_fingerprints(model, x, batch=16)
    assert model.training and z.shape == (16, 128)
    # own kNN (no teacher forcing): a near-tie flip moves one clip's z by up to ~3e-3 (tests/test_oracle_golden.py)
    assert maxerr(z[:8], g.t("z_i_eval")) < 1e-2 and maxerr(z[8:], g.t("z_j_eval")) < 1e-2
    ref = torch.cat([g.t("z_i_eval"), g.t("z_j_eval")])
    cos = torch.nn.functional.cosine_similarity(z.cpu(), ref, dim=1)
    assert float(cos.min()) > 0.9995 and float(cos.median()) > 0.999999
    z5 = extract_fingerprints(model, x, batch=5)         # ragged splits change nothing in eval mode
    assert maxerr(z5, z) < 1e-2
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(100000, 7, 8) == (87500, 100000)


def test_graph_encoder_dgl_signature_shim(golden):
    """train.py:111 / test_fp.py:235 / downstream.py:112 call sites: GraphEncoderDGL(cfg=..., in_channels=..., k=...)"""
    from neuralsampleid_amd.encoder.dgl.graph_encoder import GraphEncoderDGL
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden("e2e_b8_k3")
    model = SimCLR(GRAFP_CFG, encoder=GraphEncoderDGL(cfg=GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t"))
    load_synth(model)
    model.eval()
    x = g.t("x_i").to(DEV)
    with torch.no_grad():
        p = model.peak_extractor(x)
        x_nodes, x_emb = model.encoder(p, return_pre_proj=True)
        emb_only = model.encoder(p)
        h_i, _, _, _ = model(x, x)
    assert x_nodes.shape == (8, 512, 32) and x_emb.shape == (8, 1024)
    assert maxerr(emb_only, x_emb) < 2e-3 and maxerr(h_i, x_emb) < 2e-3
    assert relerr(x_emb, g.t("h_i_eval")) < 1e-2       # the gcn_lib encoder's numbers (own kNN: one near-tie flip)
    # proj + node mean of the returned node matrix reproduces the embedding (reference :136-140)
    w = model.encoder.proj.weight.reshape(1024, 512)
    ref = (x_nodes.mean(dim=2) @ w.t()) + model.encoder.proj.bias
    assert maxerr(ref, x_emb) < 1e-3


def test_two_stream_views_equal_sequential(golden):
    """SimCLR(overlap_views=True) runs view j on a side stream: same loss, gradients and BN running statistics as the
    sequential order (running stats must see view i first, then view j — reference simclr.py:36,42)"""
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    g = golden("e2e_b8_k3")
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    res = {}
    for overlap in (False, True):
        model = build_model(3).train()
        model.overlap_views = overlap
        opt = FusedClipAdam(model.parameters(), lr=8e-5)
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        opt.step()                                   # joins the side stream before reading the gradients
        res[overlap] = (float(loss.detach()), opt.flat_g.clone(), {k: v.clone() for k, v in model.state_dict().items()
                                                          if k.endswith(("running_mean", "running_var",
                                                                         "num_batches_tracked"))})
    assert abs(res[True][0] - res[False][0]) < 1e-5
    assert relerr(res[True][1], res[False][1]) < 1e-3                      # atomics order + kNN near-ties only
    for k, v in res[False][2].items():              # two streams: the updates are deferred to one launch after both views
        if k.endswith("num_batches_tracked"):
            assert int(res[True][2][k]) == int(v) == 2, k
        else:
            assert maxerr(res[True][2][k], v) < 1e-6, k


def test_fingerprint_db_files(tmp_path):
    """fpdb.build_fp_db: fingerprints from the HIP path land in the reference's DB format (test_fp.py:120-133); two ranks
    writing their row ranges into the shared memmap give the same file as one rank"""
    from neuralsampleid_amd import fpdb
    from neuralsampleid_amd.fingerprint import extract_fingerprints
    from oracle import ref_fpdb
    model = build_model(3).eval()
    x, _ = synth_clips(11)
    x = x.to(DEV)
    songs = [("a", x[:4]), ("b", x[4:5]), ("c", x[5:11])]
    n, d = fpdb.build_fp_db(model, songs, str(tmp_path / "one"), "ref_db", batch=4)
    ref = extract_fingerprints(model, x, 16).cpu().numpy()
    data, shape = ref_fpdb.load_memmap_data(str(tmp_path / "one"), "ref_db")      # the reference's reader
    assert (n, d) == (11, 128) and tuple(shape) == (11, 128)
    assert np.abs(np.asarray(data) - ref).max() < 1e-5
    assert fpdb.load_lookup(str(tmp_path / "one"), "ref_db") == ["a"] * 4 + ["b"] + ["c"] * 6
    for r in range(2):
        fpdb.build_fp_db(model, songs, str(tmp_path / "two"), "query_db", query_style=True, batch=4, rank=r, world=2,
                         barrier=lambda: None)
    data2, _ = ref_fpdb.load_memmap_data(str(tmp_path / "two"), "query_db")
    assert np.abs(np.asarray(data2) - ref).max() < 1e-5
    assert fpdb.load_lookup(str(tmp_path / "two"), "query_db")[4] == "b_1"


def test_graphed_train_step_equals_eager():
    """graphs.GraphedTrainStep: ONE replay of the captured step from state S = ONE eager step from state S, for the states
    S0 (initial), S1, S2 of an eager run, with new inputs copied into the static buffers each time. (Comparing whole
    trajectories instead is chaotic at B = 16: Adam's first updates are lr * sign(g), fp32 atomics order flips the sign of
    near-zero gradients, and two runs are 0.4 apart in loss two updates later — that says nothing about the capture.)"""
    from neuralsampleid_amd.graphs import GraphedTrainStep
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    x_i, x_j = (t.to(DEV) for t in synth_clips(16))
    lr = 8e-5

    def state(model, opt):
        return ([t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_count)],
                [b.clone() for b in model.buffers()])

    def load(model, opt, st):
        with torch.no_grad():
            for dst, src in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_count), st[0]):
                dst.copy_(src)
            for b, v in zip(model.buffers(), st[1]):
                b.copy_(v)

    model = build_model(3).train()
    opt = FusedClipAdam(model.parameters(), lr=lr)
    states, losses = [state(model, opt)], []
    for s_ in range(3):
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i.roll(s_, 0), x_j.roll(s_, 0))
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        states.append(state(model, opt))
    model2 = build_model(3).train()
    opt2 = FusedClipAdam(model2.parameters(), lr=lr)
    step = GraphedTrainStep(model2, opt2, GRAFP_CFG, x_i, x_j, loss_fn=ntxent_loss, warmup=1)
    assert int(opt2.step_count) == 0 and torch.equal(opt2.flat_p, states[0][0][0])     # construction restored the state
    worst_loss = worst_dp = worst_bn = 0.0
    for s_ in range(3):
        load(model2, opt2, states[s_])
        got = float(step(x_i.roll(s_, 0), x_j.roll(s_, 0)))
        torch.cuda.synchronize()
        assert int(opt2.step_count) == s_ + 1
        worst_loss = max(worst_loss, abs(got - losses[s_]))
        want = states[s_ + 1]
        worst_dp = max(worst_dp, float((opt2.flat_p - want[0][0]).abs().mean()) / lr)
        worst_bn = max(worst_bn, max(maxerr(b.double(), v.double()) / max(1.0, float(v.double().abs().max()))
                                     for b, v in zip(model2.buffers(), want[1])))
    print(f"graphed vs eager, per state: |dloss| {worst_loss:.2e}, mean |dp|/lr {worst_dp:.3e}, BN buffers {worst_bn:.2e}")
    # measured on MI355X (round 2, gpurun_out/r02_plug/graphed.log): |dloss| 1.2e-7, mean |dp|/lr 2.1e-6, BN buffers 0
    # (same state in => same forward; the update differs only where the order of fp32 atomics in the split-K GEMMs flips
    # the last bit of a near-zero gradient under Adam's normalisation). GRAPH_TOL is <= 3x those, 1e-7 floor for the buffers.
    assert worst_loss < GRAPH_TOL["loss"] and worst_dp < GRAPH_TOL["dp"] and worst_bn < GRAPH_TOL["bn"]


GRAPH_TOL = {"loss": 2e-6, "dp": 6e-6, "bn": 1e-7}      # loss: 4 ulp of a float32 near 4.6 (one ulp is 4.8e-7; measured 1-3 ulp)


def test_graphed_fingerprinter_equals_eager_extraction(golden):
    """fingerprint.GraphedFingerprinter: full micro-batches as hipGraph replays + an eager ragged tail == extract_fingerprints;
    a weight change after the capture is refused instead of silently fingerprinting with the old weights"""
    from neuralsampleid_amd.fingerprint import GraphedFingerprinter, extract_fingerprints
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    g = golden("e2e_b8_k3")
    model = build_model(3).train()
    x = torch.cat([g.t("x_i"), g.t("x_j"), g.t("x_i")[:3]]).to(DEV)          # 19 clips: 4 full micro-batches of 4 + a tail of 3
    ref = extract_fingerprints(model, x, batch=4)
    fp = GraphedFingerprinter(model, micro_batch=4)
    z = fp(x)
    assert model.training and z.shape == (19, 128)
    assert maxerr(z, ref) < 2e-6                                             # same kernels; split-K atomics in the projector
    # three concurrent replays on three HIP streams (micro-batches dealt round-robin): the same numbers, twice in a row (the static
    # buffers of a stream are reused by its next micro-batch)
    from neuralsampleid_amd import fingerprint as FP_
    fp3 = GraphedFingerprinter(model, micro_batch=4, streams=3)
    try:
        for caller_lane in (True, False):                                    # the caller's stream as one of the lanes, or idle
            FP_.USE_CALLER_STREAM = caller_lane
            for _ in range(2):
                z3 = fp3(x)
                torch.cuda.synchronize()
                assert maxerr(z3, ref) < 2e-6
    finally:
        FP_.USE_CALLER_STREAM = True
    del fp3
    assert maxerr(z[:8], g.t("z_i_eval")) < 1e-2                             # and the reference's eval goldens (own kNN)
    # torch-side weight changes move only `_version`: load_state_dict after the capture must be refused as well,
    # and a fresh capture must see the loaded weights
    sd = {k_: (v + 0.01 * torch.randn_like(v) if v.dtype.is_floating_point and "relative_pos" not in k_ else v)
          for k_, v in model.state_dict().items()}
    model.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        fp(x)
    ref2 = extract_fingerprints(model, x, batch=4)
    assert maxerr(ref2, ref) > 1e-4                                          # the perturbation is visible
    fp2 = GraphedFingerprinter(model, micro_batch=4)
    assert maxerr(fp2(x), ref2) < 2e-6
    # a parameter REPLACED in its module leaves the captured tensor untouched (same pointer, same version, no epoch bump): the guard
    # compares identities too -- load_state_dict(assign=True), attribute assignment, a swapped sub-module
    fp4 = GraphedFingerprinter(model, micro_batch=4)
    keep = {k_: v for k_, v in model.state_dict().items()}
    model.load_state_dict({k_: v.clone() for k_, v in keep.items()}, assign=True)
    with pytest.raises(RuntimeError):
        fp4(x)
    fp4 = GraphedFingerprinter(model, micro_batch=4)
    old_w = model.projector[2].weight
    model.projector[2].weight = torch.nn.Parameter(old_w.detach().clone() * 1.5)
    with pytest.raises(RuntimeError):
        fp4(x)
    model.projector[2].weight = old_w
    fp4(x)                                                                   # the captured tensor is back in place: accepted again
    old_m = model.projector[2]
    model.projector[2] = torch.nn.Linear(old_m.in_features, old_m.out_features).to(DEV)
    with pytest.raises(RuntimeError):
        fp4(x)
    model.projector[2] = old_m
    del fp4
    fp2 = GraphedFingerprinter(model, micro_batch=4)
    opt = FusedClipAdam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    _, _, z_i, z_j = model(x[:8], x[8:16])
    ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
    opt.step()
    with pytest.raises(RuntimeError):
        fp2(x)


def test_simclr_e2e_size_s(golden):
    """A second encoder size end to end (VERDICT r4 task 5): GraphEncoder(size='s') = channels 80 / 160 / 400 / 640
    (encoder/graph_encoder.py:121-123) -- none of them a shape the fused / weight-stationary kernels are specialised for, so every layer
    takes the generic tile kernels, the strip / rank kNN forms and the plain aggregation. Against the reference's own outputs
    (make_golden.py::gold_e2e_s): eval embeddings with the HIP kNN's own neighbour sets checked, then step 0 of train.py:53-75."""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden("e2e_b8_s_k3")
    with open(os.path.join(GOLDEN, "e2e_b8_s_k3_checksums.json")) as f:
        chk = json.load(f)
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="s"))
    assert {k_: list(v.shape) for k_, v in model.state_dict().items()} == chk["state_shapes"]        # the reference's key set and shapes
    model = load_synth(model).to(DEV)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    pc = lambda t: torch.stack([t.detach().double().sum(1), t.detach().double().norm(dim=1)], 1).cpu()
    gold_idx, gaps = tape_of(g, "eval")
    model.eval()
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    rec = F_.TAPE.recorded
    F_.TAPE = None
    assert len(rec) == 24
    hard, soft, rows = check_tape(rec, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    assert (pc(h_i) - g.t("h_i_eval_pc")).abs().max() < 2e-3 and (pc(h_j) - g.t("h_j_eval_pc")).abs().max() < 2e-3
    assert maxerr(z_i, g.t("z_i_eval")) < 1e-5 and maxerr(z_j, g.t("z_j_eval")) < 1e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5
    # step 0
    model.train()
    model.zero_grad()
    gold_idx, gaps = tape_of(g, "s0")
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    rec = F_.TAPE.recorded
    F_.TAPE = None
    hard, soft, rows = check_tape(rec, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    assert maxerr(z_i, g.t("z_i_train")) < 2e-5 and abs(float(loss.detach()) - float(g["loss_train"][0])) < 5e-5
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    for name in [n for n in g if n.startswith("grad.")]:
        late = name.startswith(("grad.encoder.backbone.14", "grad.encoder.proj", "grad.projector"))
        assert relerr(grads[name[5:]], g.t(name)) < (5e-4 if late else 4e-2), (name, relerr(grads[name[5:]], g.t(name)))
    worst = max(abs(float(grads[n].double().norm()) - nrm) / nrm for n, (s_, nrm) in chk["grad"].items() if nrm > 1e-3)
    assert worst < 2e-2, worst
    sd = model.state_dict()
    for name, (s_, nrm) in chk["bn_after_step1"].items():
        assert abs(float(sd[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    assert abs(gn - float(g["gnorm"][0])) / float(g["gnorm"][0]) < 5e-3


@pytest.mark.parametrize("size", ["m", "b"])
def test_simclr_e2e_sizes_m_and_default(golden, size):
    """The remaining encoder sizes of encoder/graph_encoder.py:124-129 end to end (VERDICT r5 task 7): 'm' = 96 / 192 / 384 / 768 channels x
    [2,2,16,2] blocks, the default ('b') = 128 / 256 / 512 / 1024 x [2,2,18,2]; B = 4, k = 3, strict fp32. Against the reference's own
    outputs (make_golden.py::gold_e2e_m / gold_e2e_b, checksum-only fixtures): eval embeddings with the HIP kNN's own neighbour sets
    checked, then step 0 of train.py:53-75 with the reference's neighbour ids forced."""
    import hashlib
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden(f"e2e_b4_{size}_k3")
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size=size))
    shapes = json.dumps({k_: list(v.shape) for k_, v in model.state_dict().items() if "relative_pos" not in k_}, sort_keys=True)
    assert hashlib.sha256(shapes.encode()).hexdigest() == bytes(g["state_shapes_sha"]).decode()      # the reference's key set and shapes
    model = load_synth(model).to(DEV)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    pc = lambda t: torch.stack([t.detach().double().sum(1), t.detach().double().norm(dim=1)], 1).cpu()
    n_calls = len([k_ for k_ in g if k_.startswith("knn.eval.")])
    gold_idx = [g.t(f"knn.eval.{c}").to(torch.int32) for c in range(n_calls)]
    gaps = [np.asarray(g[f"gap.eval.{c}"], dtype=np.float32) for c in range(n_calls)]
    model.eval()
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    rec = F_.TAPE.recorded
    F_.TAPE = None
    assert len(rec) == n_calls == 2 * sum(model.encoder.blocks)
    hard, soft, rows = check_tape(rec, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 3e-3, (hard, soft, rows)        # (margins are stored as fp16: the near-tie band is as wide as before)
    # per-clip [sum, norm] of h: the embeddings of these sizes reach |h| ~ 5e4 per clip: 5e-5 of the norm (measured 1.4e-5 on the sums)
    tol_pc = 5e-5 * float(g.t("h_i_eval_pc")[:, 1].max())
    assert (pc(h_i) - g.t("h_i_eval_pc")).abs().max() < tol_pc and (pc(h_j) - g.t("h_j_eval_pc")).abs().max() < tol_pc
    assert maxerr(z_i, g.t("z_i_eval")) < 1e-5 and maxerr(z_j, g.t("z_j_eval")) < 1e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5
    # step 0
    model.train()
    model.zero_grad()
    F_.TAPE = F_.KnnTape(replay=[g.t(f"knn.s0.{c}").to(torch.int32) for c in range(n_calls)])
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    F_.TAPE = None
    assert maxerr(z_i, g.t("z_i_train")) < 2e-5 and abs(float(loss.detach()) - float(g["loss_train"][0])) < 5e-5
    assert (pc(h_i) - g.t("h_i_train_pc")).abs().max() < 5e-5 * float(g.t("h_i_train_pc")[:, 1].max())
    gn_ref = np.asarray(g["grad_norms"])
    params = list(model.named_parameters())
    assert len(params) == len(gn_ref)
    worst = max((abs(float(p.grad.double().norm()) - nrm) / nrm, n) for (n, p), nrm in zip(params, gn_ref) if nrm > 1e-3)
    assert worst[0] < 2e-2, worst
    bn_ref = np.asarray(g["bn_norms_after_step1"])
    bn = [float(t.double().norm()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var"))]
    assert len(bn) == len(bn_ref) and max(abs(a - b) / max(b, 1.0) for a, b in zip(bn, bn_ref)) <= 1e-4
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    assert abs(gn - float(g["gnorm"][0])) / float(g["gnorm"][0]) < 5e-3


@pytest.mark.parametrize("size", ["m", "b"])
def test_sizes_m_and_default_in_the_timed_arithmetic(golden, size):
    """the same two models with bf16 storage: the channel counts 96 / 192 / 384 / 768 / 1024 are outside every shape table of the
    weight-stationary and fused kernels, which must decline (launch counters) and leave the generic tile kernels to run; eval
    embeddings within cos 0.999 of the reference's fp32 ones, a training step with deferred grouped weight gradients runs"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden(f"e2e_b4_{size}_k3")
    model = load_synth(SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size=size))).to(DEV)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    n_calls = len([k_ for k_ in g if k_.startswith("knn.eval.")])
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    try:
        model.eval()
        F_.TAPE = F_.KnnTape(replay=[g.t(f"knn.eval.{c}").to(torch.int32) for c in range(n_calls)])
        with torch.no_grad():
            _, _, z_i, z_j = model(x_i, x_j)
        F_.TAPE = None
        cos = torch.nn.functional.cosine_similarity(torch.cat([z_i, z_j]).float().cpu(), torch.cat([g.t("z_i_eval"), g.t("z_j_eval")]), dim=1)
        assert float(cos.min()) > 0.999, float(cos.min())
        model.train()
        opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
        ops.launch_counters(reset=True)
        F_.TAPE = F_.KnnTape(replay=[g.t(f"knn.s0.{c}").to(torch.int32) for c in range(n_calls)])
        opt.zero_grad()
        _, _, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        F_.TAPE = None
        cnt = ops.launch_counters()
        opt.step()
        torch.cuda.synchronize()
        assert cnt["wgrad_grouped"] > 0 and cnt["gemm_fwd"] > 0, cnt
        if size == "m":          # 96 / 192 / 384 / 768: no layer matches a weight-stationary shape
            assert cnt["ws_fwd"] == 0 and cnt["ws_bwd_data"] == 0, cnt
        # (4 clips under 44-48 train-mode BatchNorms: the bf16 step is far from the fp32 one -- measured |dloss| 0.61 at size 'b';
        #  the statement here is that every kernel family accepts these channel counts, not parity: that is the fp32 test above)
        assert math.isfinite(float(loss.detach())) and abs(float(loss.detach()) - float(g["loss_train"][0])) < 1.0 and float(opt.grad_norm) > 0
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.requires_grad)
    finally:
        F_.TAPE = None
        ops.set_gemm_precision("fp32")
        F_.set_activation_dtype("fp32")


def test_size_s_bf16_storage_runs_the_generic_kernels(golden):
    """the same model in the timed arithmetic (bf16 storage): eval embeddings stay within cos 0.999 of the reference's fp32 ones and a
    training step runs (finite loss, every parameter receives a gradient) -- the C = 80 / 160 / 400 / 640 shapes on the generic bf16 paths"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    from neuralsampleid_amd.simclr.simclr import SimCLR
    g = golden("e2e_b8_s_k3")
    model = load_synth(SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="s"))).to(DEV)
    x_i, x_j = g.t("x_i").to(DEV), g.t("x_j").to(DEV)
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    try:
        model.eval()
        gold_idx, _ = tape_of(g, "eval")
        F_.TAPE = F_.KnnTape(replay=gold_idx)
        with torch.no_grad():
            _, _, z_i, z_j = model(x_i, x_j)
        F_.TAPE = None
        cos = torch.nn.functional.cosine_similarity(torch.cat([z_i, z_j]).float().cpu(), torch.cat([g.t("z_i_eval"), g.t("z_j_eval")]), dim=1)
        assert float(cos.min()) > 0.999, float(cos.min())
        model.train()
        model.zero_grad()
        _, _, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        assert torch.isfinite(loss.detach()).item()
        missing = [n for n, p in model.named_parameters() if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all())]
        assert not missing, missing[:5]
    finally:
        F_.TAPE = None
        ops.set_gemm_precision("fp32")
        F_.set_activation_dtype("fp32")
