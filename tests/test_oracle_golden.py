"""Pins the oracle (oracle/ref_torch.py) to golden vectors produced by the reference's own modules."""
import json
import os

import numpy as np
import pytest
import torch

import tapes
from compare import absmax, allclose, l2norm, relerr
from conftest import GOLDEN, from_rows, to_rows
from oracle import ref_torch as R
from synth import GRAFP_CFG, synth_state, synth_tensor

torch.set_num_threads(8)


def synth_P(shapes: dict, prefix=""):
    return {k: synth_tensor(prefix + k, torch.empty(s)) for k, s in shapes.items()}


def block_shapes(C, pre=""):
    s = {}
    def bn(p, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[p + n] = (c,)
        s[p + "num_batches_tracked"] = ()
    s[pre + "0.fc1.0.weight"] = (C, C, 1, 1); s[pre + "0.fc1.0.bias"] = (C,); bn(pre + "0.fc1.1.", C)
    s[pre + "0.graph_conv.gconv.nn.0.weight"] = (2 * C, C // 2, 1, 1); s[pre + "0.graph_conv.gconv.nn.0.bias"] = (2 * C,)
    bn(pre + "0.graph_conv.gconv.nn.1.", 2 * C)
    s[pre + "0.fc2.0.weight"] = (C, 2 * C, 1, 1); s[pre + "0.fc2.0.bias"] = (C,); bn(pre + "0.fc2.1.", C)
    s[pre + "1.fc1.0.weight"] = (4 * C, C, 1, 1); bn(pre + "1.fc1.1.", 4 * C)
    s[pre + "1.fc2.0.weight"] = (C, 4 * C, 1, 1); bn(pre + "1.fc2.1.", C)
    return s


def knn_set_mismatch(idx, gold_idx, gap, tol=1e-4):
    """rows whose neighbour SET differs, excluding rows inside the oracle's near-tie margin"""
    a = np.sort(np.asarray(idx), axis=-1)
    b = np.sort(np.asarray(gold_idx), axis=-1)
    bad = (a != b).any(axis=-1)
    return int((bad & (np.asarray(gap) >= tol)).sum()), int(bad.sum())


KNN_CASES = [("c64n256", [(3, 1), (5, 1), (4, 2), (18, 3), (18, 1)]),       # k = 18: BASELINE config 4's four shapes
             ("c128n128", [(3, 1), (18, 2)]),
             ("c256n64", [(3, 1), (18, 3), (9, 2)]),
             ("c512n32", [(3, 1), (5, 2), (18, 1)]),
             ("c80n256", [(3, 1), (9, 2)])]


@pytest.mark.parametrize("tag,kds", KNN_CASES)
def test_knn(golden, tag, kds):
    g = golden("knn_" + tag)
    y = to_rows(g.t("x"))
    for k, d in kds:
        idx = R.knn_graph(y, k, d)
        gap = g[f"mingap_k{k}_d{d}"] if d > 1 else g[f"setgap_k{k}_d{d}"]
        hard, soft = knn_set_mismatch(idx, g[f"idx_k{k}_d{d}"], gap)
        assert hard == 0, (k, d, hard, soft)
        assert (idx[..., 0] == torch.arange(y.shape[1])).all()           # self is rank 0
        assert (g[f"center_k{k}_d{d}"] == np.arange(y.shape[1])[None, :, None]).all()


def test_mr_aggregate(golden):
    g = golden("mragg_c64n256")
    y = to_rows(g.t("x")).requires_grad_(True)
    u = R.mr_aggregate(y, g.t("idx").long())
    gu = to_rows(g.t("gu"))
    (u * gu).sum().backward()
    assert torch.equal(from_rows(u.detach()), g.t("u"))
    assert allclose(from_rows(y.grad), g.t("dx"), atol=1e-6)


def test_mrconv(golden):
    g = golden("mrconv_c64n256")
    C = 64
    P = synth_P({"nn.0.weight": (2 * C, C // 2, 1, 1), "nn.0.bias": (2 * C,), "nn.1.weight": (2 * C,), "nn.1.bias": (2 * C,),
                 "nn.1.running_mean": (2 * C,), "nn.1.running_var": (2 * C,)}, "mr.")
    for k in ("nn.0.weight", "nn.0.bias", "nn.1.weight", "nn.1.bias"):
        P[k].requires_grad_(True)
    y = to_rows(g.t("x")).requires_grad_(True)
    st = R.BNState()
    u = R.mr_aggregate(y, g.t("idx").long())
    B, N, _ = u.shape
    out = torch.relu(R.batchnorm_rows(R.grouped_linear(u.reshape(B * N, -1), P, "nn.0."), P, "nn.1.", True, st))
    out.backward(to_rows(g.t("gout")).reshape(B * N, -1))
    assert allclose(from_rows(out.detach().reshape(B, N, -1)), g.t("y"), atol=2e-5)
    assert allclose(from_rows(y.grad), g.t("dx"), atol=2e-5)
    assert allclose(P["nn.0.weight"].grad, g.t("dweight"), atol=2e-4)
    assert torch.allclose(P["nn.1.weight"].grad, g.t("dgamma"), atol=2e-4)
    assert torch.allclose(P["nn.1.bias"].grad, g.t("dbeta"), atol=2e-4)
    assert torch.allclose(st.updates["nn.1.running_mean"], g.t("post.nn.1.running_mean"), atol=1e-6)
    assert torch.allclose(st.updates["nn.1.running_var"], g.t("post.nn.1.running_var"), atol=1e-6)


BLOCKS = [("c64n256_k3d1", 64, 3, 1), ("c64n256_k4d2", 64, 4, 2), ("c128n128_k5d1", 128, 5, 1),
          ("c512n32_k3d1", 512, 3, 1), ("c64n256_k18d3", 64, 18, 3), ("c256n64_k18d3", 256, 18, 3)]


@pytest.mark.parametrize("tag,C,k,d", BLOCKS)
def test_block(golden, tag, C, k, d):
    g = golden("block_" + tag)
    P = synth_P(block_shapes(C), "blk.")
    x = to_rows(g.t("x"))
    y_eval = R.ffn(R.grapher(x, P, "0.", k, d, False, None), P, "1.", False, None)
    assert allclose(from_rows(y_eval), g.t("y_eval"), atol=1e-4, rtol=1e-4)
    train_keys = R.trainable_keys(P)
    for key in train_keys:
        P[key].requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    st = R.BNState()
    y = R.ffn(R.grapher(xg, P, "0.", k, d, True, st), P, "1.", True, st)
    y.backward(to_rows(g.t("gout")))
    assert allclose(from_rows(y.detach()), g.t("y_train"), atol=1e-4, rtol=1e-4)
    assert allclose(from_rows(xg.grad), g.t("dx"), atol=2e-4, rtol=1e-3)
    for key in train_keys:
        if "grad." + key in g:
            ref = g.t("grad." + key)
            assert allclose(P[key].grad, ref, atol=1e-3 * max(1.0, absmax(ref)), rtol=1e-3), key
        elif "gradchk." + key in g:
            ref = g.t("gradsample." + key)
            assert torch.allclose(P[key].grad.flatten()[::997], ref, atol=1e-3 * max(1.0, float(ref.abs().max()))), key
    for key, v in st.updates.items():
        assert torch.allclose(v.to(torch.float64), g.t("post." + key).to(torch.float64), atol=1e-5), key


def test_downsample(golden):
    g = golden("downsample_c64n256")
    C = 64
    P = synth_P({"conv.0.weight": (2 * C, C, 3, 3), "conv.0.bias": (2 * C,), "conv.1.weight": (2 * C,), "conv.1.bias": (2 * C,),
                 "conv.1.running_mean": (2 * C,), "conv.1.running_var": (2 * C,)}, "ds.")
    x = to_rows(g.t("x"))
    assert allclose(from_rows(R.downsample(x, P, "", False, None)), g.t("y_eval"), atol=2e-5)
    for key in ("conv.0.weight", "conv.0.bias", "conv.1.weight", "conv.1.bias"):
        P[key].requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    st = R.BNState()
    y = R.downsample(xg, P, "", True, st)
    y.backward(to_rows(g.t("gout")))
    assert allclose(from_rows(y.detach()), g.t("y_train"), atol=2e-5)
    assert allclose(from_rows(xg.grad), g.t("dx"), atol=2e-5)
    assert allclose(P["conv.0.weight"].grad, g.t("dweight"), atol=2e-4)
    assert torch.allclose(P["conv.1.weight"].grad, g.t("dgamma"), atol=2e-4)
    assert torch.allclose(st.updates["conv.1.running_var"], g.t("post.conv.1.running_var"), atol=1e-6)


def test_peak(golden):
    g = golden("peak_b8")
    P = synth_P({"peak_extractor.convs.0.weight": (8, 3, 4, 8), "peak_extractor.convs.0.bias": (8,)})
    for v in P.values():
        v.requires_grad_(True)
    y = R.peak_patchify(g.t("x"), P, "peak_extractor.", GRAFP_CFG)          # (B, 256, 8)
    y.backward(g.t("gout").transpose(1, 2))
    assert allclose(y.detach().transpose(1, 2), g.t("y"), atol=1e-5)
    assert allclose(P["peak_extractor.convs.0.weight"].grad, g.t("dweight"), atol=1e-3, rtol=1e-4)
    assert torch.allclose(P["peak_extractor.convs.0.bias"].grad, g.t("dbias"), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("B", [2, 8, 256])
def test_ntxent(golden, B):
    g = golden(f"ntxent_b{B}")
    zi = g.t("z_i").requires_grad_(True)
    zj = g.t("z_j").requires_grad_(True)
    loss = R.ntxent(zi, zj, float(g["tau"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 2e-6
    assert allclose(zi.grad, g.t("dz_i"), atol=1e-6)
    assert allclose(zj.grad, g.t("dz_j"), atol=1e-6)
    # sharded form: per-rank row sums add up to the global mean
    z = torch.stack((g.t("z_i"), g.t("z_j")), 1).reshape(2 * B, -1)
    half = B  # two "ranks"
    tot = R.ntxent_rows(z, 0, half, float(g["tau"])) + R.ntxent_rows(z, half, 2 * B - half, float(g["tau"]))
    assert abs(float(tot) / (2 * B) - float(g["loss"][0])) < 2e-6


def full_state_shapes(k):
    with open(os.path.join(GOLDEN, "state_shapes.json")) as f:
        return {k_: tuple(v) for k_, v in json.load(f).items()}


def tape_of(g, tag):
    n = len([k_ for k_ in g if k_.startswith(f"knn.{tag}.")])
    return [g.t(f"knn.{tag}.{c}") for c in range(n)], [g[f"gap.{tag}.{c}"] for c in range(n)]


def check_tape(recorded, gold_idx, gaps):
    hard = soft = rows = 0
    for r, gi, gp in zip(recorded, gold_idx, gaps):
        h, s_ = knn_set_mismatch(r, gi, gp)
        hard, soft, rows = hard + h, soft + s_, rows + gp.size
    return hard, soft, rows


@pytest.mark.parametrize("k", [3, 5])
def test_e2e(golden, k):
    g = golden(f"e2e_b8_k{k}")
    with open(os.path.join(GOLDEN, f"e2e_b8_k{k}_checksums.json")) as f:
        chk = json.load(f)
    P = synth_P(full_state_shapes(k))
    plan = R.encoder_plan("t", k)
    x_i, x_j = g.t("x_i"), g.t("x_j")
    tau = GRAFP_CFG["tau"]

    # --- eval mode, neighbour indices forced to the reference's
    gold_idx, gaps = tape_of(g, "eval")
    R.TAPE = R.KnnTape(replay=gold_idx)
    try:
        with torch.no_grad():
            h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, False)
            loss = R.ntxent(z_i, z_j, tau)
        recorded = R.TAPE.recorded
    finally:
        R.TAPE = None
    assert len(recorded) == len(gold_idx) == 24
    hard, soft, rows = check_tape(recorded, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    for got, name in ((h_i, "h_i_eval"), (h_j, "h_j_eval")):
        assert (got - g.t(name)).abs().max() < 1e-4 * max(1.0, float(g.t(name).abs().max())), name
    assert (z_i - g.t("z_i_eval")).abs().max() < 1e-5 and (z_j - g.t("z_j_eval")).abs().max() < 1e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5

    # --- three training steps (train.py:53-75), indices forced per step
    opt = R.AdamState({k_: P[k_] for k_ in R.trainable_keys(P)}, lr=GRAFP_CFG["lr"])
    for step in range(3):
        gold_idx, gaps = tape_of(g, f"s{step}")
        R.TAPE = R.KnnTape(replay=gold_idx)
        try:
            if step == 0:   # inspect the first forward/backward in detail
                keys = R.trainable_keys(P)
                for k_ in keys:
                    P[k_].requires_grad_(True)
                st = R.BNState()
                h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, True, st)
                R.ntxent(z_i, z_j, tau).backward()
                hard, soft, rows = check_tape(R.TAPE.recorded, gold_idx, gaps)
                assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
                assert (h_i - g.t("h_i_train")).abs().max() < 2e-4
                assert (h_j - g.t("h_j_train")).abs().max() < 2e-4
                assert (z_i - g.t("z_i_train")).abs().max() < 2e-5
                # fp32 noise floor: at B=8 the reference's own early-layer gradients sit 1.0-1.6 % (relative L2)
                # from an fp64 evaluation of the same graph (train-mode BN backward cancels heavily); late layers
                # agree to 4e-5.  Tolerances are 2.5x that floor.
                for name in [n for n in g if n.startswith("grad.")]:
                    ref = g.t(name)
                    got = P[name[5:]].grad
                    if float(ref.norm()) < 1e-5:        # conv bias in front of BN: analytically zero
                        assert float(got.norm()) < 1e-5, name
                        continue
                    late = name.startswith(("grad.encoder.backbone.14", "grad.encoder.proj", "grad.projector"))
                    rel = float((got - ref).norm() / ref.norm())
                    assert rel < (5e-4 if late else 4e-2), (name, rel)
                worst = 0.0
                for name, (s_, nrm) in chk["grad"].items():
                    if nrm > 1e-3:
                        worst = max(worst, abs(float(P[name].grad.double().norm()) - nrm) / nrm)
                assert worst < 2e-2, worst
                for name, (s_, nrm) in chk["bn_after_step1"].items():
                    assert abs(float(st.updates[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name
                for k_ in keys:
                    P[k_].requires_grad_(False)
                    P[k_].grad = None
                R.TAPE = R.KnnTape(replay=gold_idx)
            l, gn = R.train_step(P, x_i, x_j, GRAFP_CFG, plan, opt)
        finally:
            R.TAPE = None
        print("step", step, "loss", l, g["losses"][step], "gnorm", gn, g["gnorms"][step])
        assert abs(l - g["losses"][step]) < (5e-5 if step == 0 else 2e-2), (step, l, g["losses"][step])
        assert abs(gn - g["gnorms"][step]) / g["gnorms"][step] < (1e-2 if step == 0 else 0.2), (step, gn)


def deep_state_shapes():
    """state_dict of SimCLR(GraphEncoder(blocks=[4,4,12,4])) from the reference's key list + the per-block shape rule"""
    with open(os.path.join(GOLDEN, "deep_b4_k18_checksums.json")) as f:
        keys = json.load(f)["keys"]
    base = full_state_shapes(3)
    shapes = {}
    chan = {}
    plan = R.encoder_plan("t", 18, blocks=[4, 4, 12, 4], use_dilation=True)
    for i, e in enumerate(plan):
        chan[i] = e[1] if e[0] == "block" else (e[1], e[2])
    for k_ in keys:
        if not k_.startswith("encoder.backbone."):
            shapes[k_] = base[k_]
            continue
        i = int(k_.split(".")[2])
        rest = k_.split(".", 3)[3]
        if isinstance(chan[i], tuple):
            ci, co = chan[i]
            shapes[k_] = {"conv.0.weight": (co, ci, 3, 3), "conv.1.num_batches_tracked": ()}.get(rest, (co,))
        else:
            C = chan[i]
            sh = block_shapes(C)
            shapes[k_] = sh[rest] if rest in sh else None
    return shapes, plan


def test_deep_config4(golden):
    """BASELINE config 4 (blocks [4,4,12,4], k = 18, dilation 1/2/3/1 by stage) against the reference's own classes
    assembled with that schedule (tests/golden/make_golden.py::deep_reference_encoder): eval forward and step 0"""
    g = golden("deep_b4_k18")
    with open(os.path.join(GOLDEN, "deep_b4_k18_checksums.json")) as f:
        chk = json.load(f)
    shapes, plan = deep_state_shapes()
    assert [tuple(r) for r in g["plan"]] == [(e[1], n, e[2], e[3]) for e, n in zip(
        [e for e in plan if e[0] == "block"], [256] * 4 + [128] * 4 + [64] * 12 + [32] * 4)]
    P = {}
    for k_, s_ in shapes.items():
        if s_ is None:                 # relative_pos: dead in the reference (torch_vertex.py:189), never read by the oracle
            continue
        P[k_] = synth_tensor(k_, torch.empty(s_))
    x_i, x_j = g.t("x_i"), g.t("x_j")
    # the reference's graphs: the oracle's own search + the fixture's near-tie rows, every clip of every build proven by its hash
    R.TAPE = tape = R.KnnTape(patch=tapes.patches_of(g, "eval"))
    try:
        with torch.no_grad():
            h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, False)
            loss = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
    finally:
        R.TAPE = None
    assert len(tape.recorded) == 48
    hard, soft, rows = tapes.check_patched(tape, g, "eval")
    assert hard == 0 and soft <= rows * 5e-3, (hard, soft, rows)
    assert (h_i - g.t("h_i_eval")).abs().max() < 1e-4 * max(1.0, float(g.t("h_i_eval").abs().max()))
    assert (z_j - g.t("z_j_eval")).abs().max() < 2e-5
    assert abs(float(loss.detach()) - float(g["loss_eval"][0])) < 1e-5
    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    R.TAPE = tape = R.KnnTape(patch=tapes.patches_of(g, "s0"))
    try:
        st = R.BNState()
        h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, True, st)
        loss = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
        loss.backward()
    finally:
        R.TAPE = None
    hard, soft, rows = tapes.check_patched(tape, g, "s0")
    assert hard == 0 and soft <= rows * 5e-3, (hard, soft, rows)
    assert (h_i - g.t("h_i_train")).abs().max() < 5e-4 and (z_i - g.t("z_i_train")).abs().max() < 5e-5
    assert abs(float(loss.detach()) - float(g["loss_train"][0])) < 5e-5
    for name in [n for n in g if n.startswith("grad.")]:
        ref, got = g.t(name), P[name[5:]].grad
        if l2norm(ref) < 1e-5:
            assert float(got.norm()) < 1e-5, name
            continue
        late = name.startswith(("grad.encoder.backbone.26", "grad.encoder.proj", "grad.projector"))
        rel = relerr(got, ref)
        print(name, rel)
        assert rel < (1e-3 if late else 8e-2), (name, rel)         # B = 4, 24 blocks: twice the depth of test_e2e's floor
    for name, (s_, nrm) in chk["bn_after_step1"].items():
        assert abs(float(st.updates[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name


def test_e2e_size_s(golden):
    """A second encoder size ('s': channels 80 / 160 / 400 / 640, encoder/graph_encoder.py:121-123) end to end against the reference's
    own outputs (tests/golden/make_golden.py::gold_e2e_s): eval embeddings and step 0 of train.py:53-75, neighbour ids forced."""
    g = golden("e2e_b8_s_k3")
    with open(os.path.join(GOLDEN, "e2e_b8_s_k3_checksums.json")) as f:
        chk = json.load(f)
    P = synth_P({k_: tuple(v) for k_, v in chk["state_shapes"].items()})
    plan = R.encoder_plan("s", 3)
    x_i, x_j = g.t("x_i"), g.t("x_j")
    tau = GRAFP_CFG["tau"]
    pc = lambda t: torch.stack([t.detach().double().sum(1), t.detach().double().norm(dim=1)], 1)
    gold_idx, gaps = tape_of(g, "eval")
    R.TAPE = R.KnnTape(replay=gold_idx)
    try:
        with torch.no_grad():
            h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, False)
            loss = R.ntxent(z_i, z_j, tau)
        recorded = R.TAPE.recorded
    finally:
        R.TAPE = None
    assert len(recorded) == len(gold_idx) == 24
    hard, soft, rows = check_tape(recorded, gold_idx, gaps)
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    assert (pc(h_i) - g.t("h_i_eval_pc")).abs().max() < 2e-3 and (pc(h_j) - g.t("h_j_eval_pc")).abs().max() < 2e-3
    assert (z_i - g.t("z_i_eval")).abs().max() < 1e-5 and (z_j - g.t("z_j_eval")).abs().max() < 1e-5
    assert abs(float(loss) - float(g["loss_eval"][0])) < 1e-5
    # step 0
    gold_idx, gaps = tape_of(g, "s0")
    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    R.TAPE = R.KnnTape(replay=gold_idx)
    try:
        st = R.BNState()
        h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, True, st)
        loss = R.ntxent(z_i, z_j, tau)
        loss.backward()
        hard, soft, rows = check_tape(R.TAPE.recorded, gold_idx, gaps)
    finally:
        R.TAPE = None
    assert hard == 0 and soft <= rows * 2e-3, (hard, soft, rows)
    assert (z_i - g.t("z_i_train")).abs().max() < 2e-5 and abs(float(loss) - float(g["loss_train"][0])) < 2e-5
    for name in [n for n in g if n.startswith("grad.")]:
        ref, got = g.t(name), P[name[5:]].grad
        late = name.startswith(("grad.encoder.backbone.14", "grad.encoder.proj", "grad.projector"))
        assert float((got - ref).norm() / ref.norm()) < (5e-4 if late else 4e-2), name
    worst = max(abs(float(P[n].grad.double().norm()) - nrm) / nrm for n, (s_, nrm) in chk["grad"].items() if nrm > 1e-3)
    assert worst < 2e-2, worst
    for name, (s_, nrm) in chk["bn_after_step1"].items():
        assert abs(float(st.updates[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name
