#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's live modules (CPU, fp32) in this container.

Run from the repo root, in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

It imports /root/reference with three stub modules in sys.modules for third-party imports that
are unused on the hot path (timm.models.layers.DropPath / to_2tuple / trunc_normal_,
torchmetrics.functional.pairwise_cosine_similarity, torchvision), feeds every module synthesized
weights (synth.py) and seeded inputs, and writes small .npz fixtures next to this file.
Only data (inputs, outputs, checksums) is written; no reference source is copied.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synth import GRAFP_CFG, fixed_graph, row_set_hash, synth_clips, synth_randn, synth_state, synth_unit_pair  # noqa: E402

REF = os.environ.get("NSID_REFERENCE", "/root/reference")


def _install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):  # timm 0.9.16 semantics; never instantiated with p>0 at default settings
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    stub("timm")
    stub("timm.models")
    stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x), trunc_normal_=None)
    stub("torchmetrics")
    stub("torchmetrics.functional", pairwise_cosine_similarity=None)
    stub("torchvision")


_install_stubs()
sys.path.insert(0, REF)
from encoder.gcn_lib.torch_edge import DenseDilatedKnnGraph  # noqa: E402
from encoder.gcn_lib.torch_vertex import Grapher, MRConv2d  # noqa: E402
from encoder.graph_encoder import FFN, Downsample, GraphEncoder  # noqa: E402
from peak_extractor import GPUPeakExtractorv2  # noqa: E402
from simclr.ntxent import ntxent_loss  # noqa: E402
from simclr.simclr import SimCLR  # noqa: E402

torch.set_num_threads(8)
CFG = dict(GRAFP_CFG)


# Inputs that a RULE reproduces (synth.synth_randn / synth_clips, bench_clips) are not stored: the fixture records which rule made them
# ("__synth__": key -> [rule, args]) and tests/conftest.py regenerates them on load — random fp32 inputs do not compress, and they were a
# third of the bytes under tests/golden ("small fixtures").
_RULE_MADE = []          # (rule name, args, tensor) of every rule call of this run
_DIGESTS = {}
_synth_randn, _synth_clips = synth_randn, synth_clips


def synth_randn(tag, *shape):      # noqa: F811  (shadows the import on purpose)
    t = _synth_randn(tag, *shape)
    _RULE_MADE.append(("randn", [tag, list(shape)], t.clone()))
    return t


def synth_clips(batch):            # noqa: F811
    x_i, x_j = _synth_clips(batch)
    _RULE_MADE.extend([("clips_i", [batch], x_i.clone()), ("clips_j", [batch], x_j.clone())])
    return x_i, x_j


def save(name, **arrays):
    out, synth = {}, {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        rule = next(((r, a) for r, a, t in _RULE_MADE if tuple(t.shape) == v.shape and v.dtype == np.float32
                     and np.array_equal(t.numpy(), v)), None)
        if rule is not None:
            synth[k] = [rule[0], rule[1]]
            _DIGESTS[f"{name}/{k}"] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16]
            continue
        out[k] = v
    if synth:
        out["__synth__"] = np.frombuffer(json.dumps(synth, sort_keys=True).encode(), dtype=np.uint8)
        # digests of the rule-made inputs (tests/conftest.py checks them on load); a partial run keeps the other fixtures' entries
        dpath = os.path.join(HERE, "synth_digests.json")
        merged = json.load(open(dpath)) if os.path.exists(dpath) else {}
        merged.update(_DIGESTS)
        json.dump(merged, open(dpath, "w"), indent=0, sort_keys=True)
    from compact import compact_fixture          # large reference tensors -> sample + checksums, margins -> bits (tests/golden/compact.py)
    out = compact_fixture(name, out)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KB" + (f"  (by rule: {sorted(synth)})" if synth else ""))


def load_synth(module, prefix=""):
    module.load_state_dict(synth_state(module.state_dict(), prefix))
    return module


def bn_stats(module):
    return {k: v.clone() for k, v in module.state_dict().items()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


# ---------------------------------------------------------------- kNN graph
def gold_knn():
    print("knn")
    # the last entries of each list are BASELINE config 4's shapes (k = 18, dilation 1 / 2 / 3 / 1 by stage)
    cases = [("c64n256", 2, 64, 256, [(3, 1), (5, 1), (4, 2), (18, 3), (18, 1)]),
             ("c128n128", 2, 128, 128, [(3, 1), (18, 2)]),
             ("c256n64", 3, 256, 64, [(3, 1), (18, 3), (9, 2)]),
             ("c512n32", 3, 512, 32, [(3, 1), (5, 2), (18, 1)]),
             ("c80n256", 2, 80, 256, [(3, 1), (9, 2)])]          # size 's' channel count: the general (strip) kernel
    for tag, B, C, N, kds in cases:
        x = synth_randn("knn_" + tag, B, C, N, 1)
        out = {"x": x}
        # the reference's own distance on its own normalised features, for the margin rule
        xn = torch.nn.functional.normalize(x, p=2.0, dim=1).transpose(2, 1).squeeze(-1)
        dist = (xn * xn).sum(-1, keepdim=True) + (-2 * xn @ xn.transpose(2, 1)) \
            + (xn * xn).sum(-1, keepdim=True).transpose(2, 1)
        dsort = dist.sort(dim=-1).values
        for k, d in kds:
            g = DenseDilatedKnnGraph(k, d, False, 0.0)
            ei = g(x)
            assert ei.shape == (2, B, N, k)
            kd = k * d
            top = dsort[..., : min(kd + 1, N)]
            gaps = top[..., 1:] - top[..., :-1]
            out[f"idx_k{k}_d{d}"] = ei[0].to(torch.int32)
            out[f"center_k{k}_d{d}"] = ei[1].to(torch.int32)
            out[f"mingap_k{k}_d{d}"] = gaps.min(dim=-1).values      # strict-order margin over ranks 0..kd
            out[f"setgap_k{k}_d{d}"] = gaps[..., -1] if kd < N else torch.full((B, N), 1.0)
        save("knn_" + tag, **out)


# ---------------------------------------------------------------- MRConv2d (gather + max-relative + grouped conv)
def gold_mrconv():
    print("mrconv")
    B, C, N, k = 2, 64, 256, 3
    x = synth_randn("mr_x", B, C, N, 1).requires_grad_(True)
    ei = DenseDilatedKnnGraph(k, 1, False, 0.0)(x.detach())
    m = load_synth(MRConv2d(C, 2 * C, "relu", "batch", True), "mr.")
    m.train()
    captured = {}
    m.nn.register_forward_pre_hook(lambda mod, inp: captured.__setitem__("u", inp[0].detach().clone()))
    y = m(x, ei)
    gout = synth_randn("mr_gout", *y.shape)
    y.backward(gout)
    save("mrconv_c64n256", x=x, idx=ei[0].to(torch.int32), u=captured["u"], y=y, gout=gout, dx=x.grad,
         dweight=m.nn[0].weight.grad, dbias=m.nn[0].bias.grad,
         dgamma=m.nn[1].weight.grad, dbeta=m.nn[1].bias.grad,
         **{"post." + k_: v for k_, v in bn_stats(m).items()})
    # aggregate-only backward: gradient of sum(u * gu) w.r.t. x, exercising the arg-max routing
    x2 = x.detach().clone().requires_grad_(True)
    m2 = MRConv2d(C, 2 * C, "relu", "batch", True)
    cap2 = {}
    m2.nn.register_forward_pre_hook(lambda mod, inp: cap2.__setitem__("u", inp[0]))
    m2(x2, ei)
    gu = synth_randn("mr_gu", B, 2 * C, N, 1)
    (cap2["u"] * gu).sum().backward()
    save("mragg_c64n256", x=x2, idx=ei[0].to(torch.int32), u=cap2["u"], gu=gu, dx=x2.grad)


# ---------------------------------------------------------------- Grapher + FFN block
def gold_block():
    print("block")
    cases = [("c64n256_k3d1", 2, 64, 256, 3, 1), ("c64n256_k4d2", 2, 64, 256, 4, 2),
             ("c128n128_k5d1", 2, 128, 128, 5, 1), ("c512n32_k3d1", 4, 512, 32, 3, 1),
             ("c64n256_k18d3", 2, 64, 256, 18, 3), ("c256n64_k18d3", 4, 256, 64, 18, 3)]
    for tag, B, C, N, k, d in cases:
        blk = nn.Sequential(
            Grapher(C, k, d, "mr", "relu", "batch", True, False, 0.2, 1, n=N, drop_path=0.0, relative_pos=True),
            FFN(in_features=C, hidden_features=4 * C, out_features=C, act="relu", drop_path=0.0))
        load_synth(blk, "blk.")
        x = synth_randn("blk_x_" + tag, B, C, N, 1)
        blk.eval()
        with torch.no_grad():
            y_eval = blk(x)
        blk.train()
        xg = x.clone().requires_grad_(True)
        y = blk(xg)
        gout = synth_randn("blk_g_" + tag, *y.shape)
        y.backward(gout)
        grads = {}
        for n_, p in blk.named_parameters():
            if p.grad is None:
                continue
            if p.grad.numel() <= 70000:
                grads["grad." + n_] = p.grad
            else:   # big weight grads: checksum + strided sample keeps the fixture small
                g64 = p.grad.double()
                grads["gradchk." + n_] = np.array([float(g64.sum()), float(g64.norm())])
                grads["gradsample." + n_] = p.grad.flatten()[::997].clone()
        save("block_" + tag, x=x, y_eval=y_eval, y_train=y, gout=gout, dx=xg.grad, **grads,
             **{"post." + k_: v for k_, v in bn_stats(blk).items()})


# ---------------------------------------------------------------- Downsample
def gold_downsample():
    print("downsample")
    B, C, N = 2, 64, 256
    ds = load_synth(Downsample(C, 2 * C), "ds.")
    x = synth_randn("ds_x", B, C, N, 1)
    ds.eval()
    with torch.no_grad():
        y_eval = ds(x)
    ds.train()
    xg = x.clone().requires_grad_(True)
    y = ds(xg)
    gout = synth_randn("ds_g", *y.shape)
    y.backward(gout)
    save("downsample_c64n256", x=x, y_eval=y_eval, y_train=y, gout=gout, dx=xg.grad,
         dweight=ds.conv[0].weight.grad, dbias=ds.conv[0].bias.grad,
         dgamma=ds.conv[1].weight.grad, dbeta=ds.conv[1].bias.grad,
         **{"post." + k_: v for k_, v in bn_stats(ds).items()})


# ---------------------------------------------------------------- peak extractor
def gold_peak():
    print("peak")
    pe = load_synth(GPUPeakExtractorv2(CFG), "peak_extractor.")
    x, _ = synth_clips(8)
    y = pe(x)
    gout = synth_randn("peak_g", *y.shape)
    y.backward(gout)
    save("peak_b8", x=x, y=y, gout=gout, dweight=pe.convs[0].weight.grad, dbias=pe.convs[0].bias.grad)


# ---------------------------------------------------------------- NT-Xent
def gold_ntxent():
    print("ntxent")
    for B in (2, 8, 256):
        zi, zj = synth_unit_pair(B)                       # (a rule: the fixture stores the outputs only)
        _RULE_MADE.extend([("unit_i", [B], zi.clone()), ("unit_j", [B], zj.clone())])
        zi, zj = zi.requires_grad_(True), zj.requires_grad_(True)
        loss = ntxent_loss(zi, zj, CFG)
        loss.backward()
        save(f"ntxent_b{B}", z_i=zi, z_j=zj, loss=loss.detach().reshape(1), dz_i=zi.grad, dz_j=zj.grad,
             tau=np.float32(CFG["tau"]))


# ---------------------------------------------------------------- end to end
def _checksums(named):
    return {n: [float(t.double().sum()), float(t.double().norm())] for n, t in named}


class KnnTape:
    """Records, for every DenseDilatedKnnGraph call of the reference model, the neighbour indices it produced
    and the oracle-side margin (k-th vs (k+1)-th distance) that decides whether a differing SET is a near-tie."""

    def __init__(self, model):
        self.idx, self.gap = [], []
        for m in model.modules():
            if isinstance(m, DenseDilatedKnnGraph):
                m.register_forward_hook(self._hook)

    def _hook(self, mod, inp, out):
        x = inp[0].detach()
        kd = mod.k * mod.dilation
        xn = torch.nn.functional.normalize(x, p=2.0, dim=1).transpose(2, 1).squeeze(-1)
        sq = (xn * xn).sum(-1, keepdim=True)
        dist = sq + (-2 * xn @ xn.transpose(2, 1)) + sq.transpose(2, 1)
        top = dist.sort(dim=-1).values[..., : min(kd + 1, dist.shape[-1])]
        gaps = top[..., 1:] - top[..., :-1]
        self.idx.append(out[0].to(torch.int16).clone())
        self.gap.append((gaps.min(-1).values if mod.dilation > 1 else gaps[..., -1]).clone())

    def take(self, tag):
        out = {}
        for c, (i, g) in enumerate(zip(self.idx, self.gap)):
            out[f"knn.{tag}.{c}"] = i
            out[f"gap.{tag}.{c}"] = g
        self.idx, self.gap = [], []
        return out


def gold_e2e():
    print("e2e")
    B = 8
    x_i, x_j = synth_clips(B)
    for k in (3, 5):
        torch.manual_seed(1234)
        model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=k, size="t"))
        load_synth(model)
        tape = KnnTape(model)
        model.eval()
        with torch.no_grad():
            h_i, h_j, z_i, z_j = model(x_i, x_j)
            loss_eval = ntxent_loss(z_i, z_j, CFG)
        ev = dict(h_i_eval=h_i, h_j_eval=h_j, z_i_eval=z_i, z_j_eval=z_j, loss_eval=loss_eval.reshape(1))
        ev.update(tape.take("eval"))

        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=CFG["lr"])
        losses, gnorms = [], []
        first = {}
        for step in range(3):       # train.py:53-75 step semantics on a fixed batch
            opt.zero_grad()
            h_i, h_j, z_i, z_j = model(x_i, x_j)
            loss = ntxent_loss(z_i, z_j, CFG)
            loss.backward()
            first.update(tape.take(f"s{step}"))
            if step == 0:
                first.update(h_i_train=h_i, h_j_train=h_j, z_i_train=z_i, z_j_train=z_j)
                sums = _checksums((n, p.grad) for n, p in model.named_parameters() if p.grad is not None)
                full = {"grad." + n: p.grad.clone() for n, p in model.named_parameters() if n in (
                    "peak_extractor.convs.0.weight", "encoder.stem.0.weight", "encoder.backbone.0.0.fc1.0.weight",
                    "encoder.backbone.0.0.graph_conv.gconv.nn.0.weight", "encoder.backbone.2.conv.0.bias",
                    "encoder.backbone.14.1.fc2.1.weight", "encoder.proj.bias", "projector.2.bias")}
                stats1 = _checksums((n, t.float()) for n, t in model.state_dict().items()
                                    if n.endswith(("running_mean", "running_var")))
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            opt.step()
            losses.append(float(loss.detach()))
            gnorms.append(float(gn))
        save(f"e2e_b8_k{k}", x_i=x_i, x_j=x_j, **ev, **first, **full,
             losses=np.array(losses, np.float64), gnorms=np.array(gnorms, np.float64))
        with open(os.path.join(HERE, f"e2e_b8_k{k}_checksums.json"), "w") as f:
            json.dump({"grad": sums, "bn_after_step1": stats1}, f, indent=0)
        print("   losses", losses, "gnorm", gnorms)


def gold_e2e_s():
    """A second encoder size end to end (VERDICT r4 task 5): size 's' = channels 80 / 160 / 400 / 640 (encoder/graph_encoder.py:121-123),
    B = 8, k = 3: eval embeddings, then step 0 of train.py:53-75; neighbour ids + margins of both passes for teacher forcing, per-clip
    embedding checksums instead of the h matrices, gradient / running-statistics checksums and three full gradients."""
    print("e2e_s")
    B, k = 8, 3
    x_i, x_j = synth_clips(B)
    torch.manual_seed(1234)
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=k, size="s"))
    load_synth(model)
    tape = KnnTape(model)
    model.eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss_eval = ntxent_loss(z_i, z_j, CFG)
    ev = dict(h_i_eval_pc=_per_clip(h_i), h_j_eval_pc=_per_clip(h_j), z_i_eval=z_i, z_j_eval=z_j, loss_eval=loss_eval.reshape(1))
    ev.update(tape.take("eval"))
    model.train()
    model.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    tr = dict(h_i_train_pc=_per_clip(h_i), h_j_train_pc=_per_clip(h_j), z_i_train=z_i, z_j_train=z_j, loss_train=loss.detach().reshape(1))
    tr.update(tape.take("s0"))
    sums = _checksums((n, p.grad) for n, p in model.named_parameters() if p.grad is not None)
    full = {"grad." + n: p.grad.clone() for n, p in model.named_parameters() if n in (
        "encoder.backbone.0.0.fc1.0.weight", "encoder.backbone.14.1.fc2.1.weight", "projector.2.bias")}
    stats1 = _checksums((n, t.float()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var")))
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    save("e2e_b8_s_k3", x_i=x_i, x_j=x_j, **ev, **tr, **full, gnorm=np.array([gn], np.float64))
    with open(os.path.join(HERE, "e2e_b8_s_k3_checksums.json"), "w") as f:
        json.dump({"grad": sums, "bn_after_step1": stats1, "state_shapes": {n: list(t.shape) for n, t in model.state_dict().items()}},
                  f, indent=0)
    print("   loss eval", float(loss_eval), "train", float(loss), "gnorm", gn)


def _gold_e2e_size(size, B=4, k=3):
    """Checksum-only golden of a further encoder size (VERDICT r5 task 7; encoder/graph_encoder.py:124-129: 'm' = 96 / 192 / 384 / 768
    x [2,2,16,2], anything else = 128 / 256 / 512 / 1024 x [2,2,18,2]): B = 4, eval embeddings and step 0 of train.py:53-75. Stored: z of
    both passes, per-clip checksums of h, the losses, neighbour ids of both passes as uint8 (N <= 256) with the eval pass's margins as
    float16 (the free-running comparison), and the gradient / running-statistics norms as vectors in named_parameters / state_dict order."""
    print("e2e_" + size)
    x_i, x_j = synth_clips(B)
    torch.manual_seed(1234)
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=k, size=size))
    load_synth(model)
    tape = KnnTape(model)
    model.eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss_eval = ntxent_loss(z_i, z_j, CFG)
    ev = dict(h_i_eval_pc=_per_clip(h_i), h_j_eval_pc=_per_clip(h_j), z_i_eval=z_i, z_j_eval=z_j, loss_eval=loss_eval.reshape(1))
    for name, t in tape.take("eval").items():
        ev[name] = t.to(torch.uint8) if name.startswith("knn.") else t.to(torch.float16)
    model.train()
    model.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    tr = dict(h_i_train_pc=_per_clip(h_i), h_j_train_pc=_per_clip(h_j), z_i_train=z_i, z_j_train=z_j, loss_train=loss.detach().reshape(1))
    for name, t in tape.take("s0").items():
        if name.startswith("knn."):
            tr[name] = t.to(torch.uint8)
    gnorms = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()], np.float64)
    bnnorms = np.array([float(t.double().norm()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var"))],
                       np.float64)
    shapes = json.dumps({n: list(t.shape) for n, t in model.state_dict().items() if "relative_pos" not in n}, sort_keys=True)
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    save(f"e2e_b{B}_{size}_k{k}", x_i=x_i, x_j=x_j, **ev, **tr, grad_norms=gnorms, bn_norms_after_step1=bnnorms, gnorm=np.array([gn], np.float64),
         state_shapes_sha=np.frombuffer(hashlib.sha256(shapes.encode()).hexdigest().encode(), dtype=np.uint8))
    print("   loss eval", float(loss_eval), "train", float(loss), "gnorm", gn, "params", sum(p.numel() for p in model.parameters()))


def gold_e2e_m():
    _gold_e2e_size("m")


def gold_e2e_b():
    _gold_e2e_size("b")


def bench_clips(batch, seed):
    """bench.py's synth_clips — the inputs of the TIMED step: seeds (seed, seed + 1)"""
    gi = torch.Generator().manual_seed(seed)
    gj = torch.Generator().manual_seed(seed + 1)
    x_i = torch.randn(batch, CFG["n_mels"], CFG["n_frames"], generator=gi) * 20.0 - 40.0
    x_j = x_i + 3.0 * torch.randn(batch, CFG["n_mels"], CFG["n_frames"], generator=gj)
    _RULE_MADE.extend([("bench_i", [batch, seed], x_i.clone()), ("bench_j", [batch, seed], x_j.clone())])
    return x_i, x_j


def _per_clip(t):
    """(B, D) -> (B, 2) float64 [sum, norm] per clip: pins every clip's embedding without storing it"""
    t = t.detach().double()
    return torch.stack([t.sum(1), t.norm(dim=1)], 1)


def _b256_run(x_i, x_j, threads=8):
    """step 0 of train.py:53-75 on the reference, seed-42 default initialisation; returns everything the fixtures need"""
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t"))
    tape = KnnTape(model)
    model.train()
    model.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    torch.set_num_threads(8)
    return model, tape, (h_i.detach(), h_j.detach(), z_i.detach(), z_j.detach()), float(loss.detach()), gn, grads


def _pack_tape(tape_dict, tag, n_calls):
    """neighbour ids of every call as uint8 (N <= 256) + the near-tie rows (k-th / (k+1)-th gap < 1e-4) as packed bits"""
    out = {}
    for c in range(n_calls):
        idx = tape_dict[f"knn.{tag}.{c}"]
        assert int(idx.max()) < 256 and int(idx.min()) >= 0
        out[f"knn.{tag}.{c}"] = idx.to(torch.uint8)
        out[f"near.{tag}.{c}"] = np.packbits((tape_dict[f"gap.{tag}.{c}"] < 1e-4).numpy().reshape(-1))
    return out


def gold_b256():
    """The TIMED configuration (BASELINE config 2 as bench.py runs it): seed-42 default initialisation, the bench's own 256 clip
    pairs (seeds 1000 / 1001), k = 3. One eval forward, then step 0 of train.py:53-75 (forward in train mode, NT-Xent, backward,
    clip_grad_norm_(1.0)), with the neighbour ids of all 24 graph builds of each pass.

    Also measured here, on the reference itself: HOW CHAOTIC the training-mode step is at this batch. VERDICT r2 expected that
    with BatchNorm averaging over 65 536 rows a flipped kNN near-tie would stop mattering; the reference says otherwise — scaling
    x_i by (1 + 1e-7) (view j untouched) flips neighbours from block 1 on (rows with identical ids per block: 1.0, 0.9999, ...,
    0.67 at block 12) and moves the loss by 2e-2, the gradient norm by 6 % and single gradients by 25-40 %, while a different
    thread count (another summation order, same graphs) moves nothing. So the numbers below pin the HIP path with the
    reference's neighbour ids teacher-forced, exactly like the B = 8 goldens, and the free-running comparison is bounded by the
    reference's own response (b256_seed42_k3_chaos.json)."""
    print("b256")
    B = 256
    x_i, x_j = bench_clips(B, 1000)
    torch.manual_seed(42)
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t"))
    tape = KnnTape(model)
    model.eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss_eval = ntxent_loss(z_i, z_j, CFG)
    ev_knn = tape.take("eval")
    out = dict(z_i_eval=z_i, z_j_eval=z_j, h_i_eval_clip=_per_clip(h_i), h_j_eval_clip=_per_clip(h_j), h_i_eval_head=h_i[:8],
               loss_eval=loss_eval.reshape(1).double())
    out["near_ties_eval"] = np.array([int((ev_knn[f"gap.eval.{c}"] < 1e-4).sum()) for c in range(24)], np.int64)
    del model, tape
    model, tape, (h_i, h_j, z_i, z_j), loss, gn, grads = _b256_run(x_i, x_j)
    tr_knn = tape.take("s0")
    out["near_ties_train"] = np.array([int((tr_knn[f"gap.s0.{c}"] < 1e-4).sum()) for c in range(24)], np.int64)
    out.update(_pack_tape(tr_knn, "s0", 24))
    out.update(z_i_train=z_i, z_j_train=z_j, h_i_train_clip=_per_clip(h_i), h_j_train_clip=_per_clip(h_j),
               h_i_train_head=h_i[:8], loss_train=np.array([loss], np.float64), gnorm=np.array([gn], np.float64))
    sums = _checksums(grads.items())
    out.update({"grad." + n: grads[n] for n in (
        "peak_extractor.convs.0.weight", "encoder.stem.0.weight", "encoder.backbone.0.0.fc1.0.weight",
        "encoder.backbone.0.0.graph_conv.gconv.nn.0.weight", "encoder.backbone.2.conv.0.bias",
        "encoder.backbone.7.1.fc1.1.weight", "encoder.backbone.14.1.fc2.1.weight", "encoder.proj.bias", "projector.2.bias")})
    stats1 = _checksums((n, t.float()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var")))
    save("b256_seed42_k3", **out)
    with open(os.path.join(HERE, "b256_seed42_k3_checksums.json"), "w") as f:
        json.dump({"grad": sums, "bn_after_step1": stats1}, f, indent=0)
    print("   loss eval", float(loss_eval), "train", loss, "gnorm", gn,
          "near ties eval/train", int(out["near_ties_eval"].sum()), int(out["near_ties_train"].sum()))

    # ---- the reference's own sensitivity at this batch (see the docstring)
    def response(other):
        _, tp, (_, _, zi, zj), l_, g_, gr = other
        rows = [float((a == b).all(-1).float().mean()) for a, b in zip(tape_idx, tp.idx)]
        rel = {n: float((gr[n] - grads[n]).norm() / grads[n].norm()) for n in (
            "encoder.stem.0.weight", "encoder.backbone.7.1.fc1.0.weight", "encoder.backbone.14.1.fc2.0.weight", "projector.2.weight")}
        flat = torch.cat([(gr[n] - grads[n]).reshape(-1) for n in grads])
        return {"dloss": abs(l_ - loss), "gnorm_rel": abs(g_ - gn) / gn, "max_dz": float(max((zi - z_i).abs().max(), (zj - z_j).abs().max())),
                "flat_grad_rel": float(flat.norm() / gn), "grad_rel": rel, "rows_with_equal_ids_per_graph_build": rows}
    tape_idx = [tr_knn[f"knn.s0.{c}"] for c in range(24)]
    chaos = {"perturbation x_i*(1+1e-7)": response(_b256_run(x_i * (1.0 + 1e-7), x_j)),
             "3 threads instead of 8 (summation order only)": response(_b256_run(x_i, x_j, threads=3))}
    with open(os.path.join(HERE, "b256_seed42_k3_chaos.json"), "w") as f:
        json.dump(chaos, f, indent=1)
    for k_, v in chaos.items():
        print("   ", k_, {a: b for a, b in v.items() if a != "rows_with_equal_ids_per_graph_build"})


def deep_reference_encoder(k=18, blocks=(4, 4, 12, 4)):
    """BASELINE config 4 built from the REFERENCE's own classes. The reference's GraphEncoder never advances its block
    counter (graph_encoder.py:161-173: every Grapher gets dilation 1) and has no [4,4,12,4] size, so the backbone is
    assembled here from the reference's Grapher / FFN / Downsample with the schedule its constructor spells out
    (`min(idx // 4 + 1, max_dilation)`), capped so that k * dilation fits the stage's node count — the plan
    oracle.ref_torch.encoder_plan(use_dilation=True) and neuralsampleid_amd GraphEncoder(use_dilation=True) follow."""
    enc = GraphEncoder(CFG, in_channels=CFG["n_filters"], k=k, size="t")
    channels = [64, 128, 256, 512]
    n_pos = n_real = CFG["n_mels"] * CFG["n_frames"] // (CFG["patch_bins"] * CFG["patch_frames"])
    max_d = max(128 // k, 1)
    layers, idx, plan = [], 0, []
    for i, nb in enumerate(blocks):
        if i > 0:
            layers.append(Downsample(channels[i - 1], channels[i]))
            n_pos //= 4
            n_real = (n_real - 1) // 2 + 1
        for _ in range(nb):
            d = max(1, min(idx // 4 + 1, max_d, n_real // k))
            plan.append((channels[i], n_real, k, d))
            layers.append(nn.Sequential(
                Grapher(channels[i], k, d, "mr", "relu", "batch", True, False, 0.2, 1, n=n_pos, drop_path=0.0,
                        relative_pos=True),
                FFN(in_features=channels[i], hidden_features=channels[i] * 4, out_features=channels[i], act="relu",
                    drop_path=0.0)))
            idx += 1
    enc.backbone = nn.Sequential(*layers)
    return enc, plan


def gold_deep():
    """config 4 (24 blocks, k = 18, dilated): eval forward and step 0 of training at B = 4"""
    print("deep")
    B = 4
    x_i, x_j = synth_clips(B)
    enc, plan = deep_reference_encoder()
    model = SimCLR(CFG, enc)
    load_synth(model)
    tape = KnnTape(model)
    model.eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss_eval = ntxent_loss(z_i, z_j, CFG)
    out = dict(x_i=x_i, x_j=x_j, h_i_eval=h_i, h_j_eval=h_j, z_i_eval=z_i, z_j_eval=z_j, loss_eval=loss_eval.reshape(1),
               plan=np.array(plan, np.int32))
    out.update(tape.take("eval"))
    model.train()
    model.zero_grad()
    h_i, h_j, z_i, z_j = model(x_i, x_j)
    loss = ntxent_loss(z_i, z_j, CFG)
    loss.backward()
    out.update(tape.take("s0"))
    out.update(h_i_train=h_i, h_j_train=h_j, z_i_train=z_i, z_j_train=z_j, loss_train=loss.detach().reshape(1))
    sums = _checksums((n, p.grad) for n, p in model.named_parameters() if p.grad is not None)
    out.update({"grad." + n: p.grad.clone() for n, p in model.named_parameters() if n in (
        "encoder.stem.0.weight", "encoder.backbone.0.0.fc1.0.weight", "encoder.backbone.12.0.graph_conv.gconv.nn.0.weight",
        "encoder.backbone.26.1.fc2.1.weight", "encoder.proj.bias", "projector.2.bias")})
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    out["gnorm"] = np.array([float(gn)])
    stats1 = _checksums((n, t.float()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var")))
    save("deep_b4_k18", **out)
    with open(os.path.join(HERE, "deep_b4_k18_checksums.json"), "w") as f:
        json.dump({"grad": sums, "bn_after_step1": stats1, "keys": list(model.state_dict().keys())}, f, indent=0)
    print("   plan", sorted(set(plan)), "loss eval", float(loss_eval), "train", float(loss), "gnorm", float(gn))


class FixedGraph:
    """forward hooks that REPLACE the output of every DenseDilatedKnnGraph of the reference model with synth.fixed_graph: the
    reference's modules then aggregate over a graph the HIP side can reproduce without storing ids"""

    def __init__(self, model):
        self.call = 0
        for m in model.modules():
            if isinstance(m, DenseDilatedKnnGraph):
                m.register_forward_hook(self._hook)

    def _hook(self, mod, inp, out):
        nn_idx, center = out[0], out[1]
        B, N, k = nn_idx.shape
        fixed = fixed_graph(B, N, k, self.call).to(nn_idx.dtype)
        self.call += 1
        return torch.stack((fixed, center), dim=0)


def _view(model, x):
    """one view of the reference's SimCLR.forward (simclr/simclr.py:31-47), so that a step can be run view by view"""
    h = model.encoder(model.peak_extractor(x))
    z = torch.nn.functional.normalize(model.projector(h), p=2)
    return h, z


def gold_deep_b256():
    """BASELINE config 4 at the TIMED batch: 24 blocks, k = 18, dilated, B = 256 (the bench's clip pairs, seeds 1000 / 1001), weights
    from the per-key rule (the reference has no [4,4,12,4] constructor, so its seeded default initialisation does not exist).
    With k * dilation = 18 ... 54 neighbours per row 45 % of all rows hold a near-tie (gap < 1e-4 somewhere in the sorted list the
    dilation walks), so a free-running comparison says nothing beyond the FIRST graph build of a view, and the reference's own ids
    would be a 22 MB fixture. Hence:
      first graph build of each view (identical features on both sides): an 8-bit hash of every row's neighbour SET + the near-tie
              mask — pins the search kernels at this batch against the reference's search;
      eval and step 0 (train.py:53-75): every graph replaced, on BOTH sides, by synth.fixed_graph -> embeddings, per-clip checksums,
              losses, gradient checksums of all parameters, six full gradients, running statistics. The backward is run view by view
              (same numbers: a k = 18 graph keeps ~0.9 GB of gather temporaries per block and view alive at this batch; both views at
              once do not fit this container)."""
    print("deep_b256")
    B = 256
    x_i, x_j = bench_clips(B, 1000)
    enc, plan = deep_reference_encoder()
    n_calls = len(plan)
    model = SimCLR(CFG, enc)
    load_synth(model)
    tape = KnnTape(model)
    model.eval()
    with torch.no_grad():
        model(x_i, x_j)
    ev = tape.take("own")
    out = dict(plan=np.array(plan, np.int32))
    for c in (0, n_calls):                         # block 0 of view i and of view j
        out[f"knnhash.own.{c}"] = row_set_hash(ev[f"knn.own.{c}"])
        out[f"near.own.{c}"] = np.packbits((ev[f"gap.own.{c}"] < 1e-4).numpy().reshape(-1))
    print("   near-tie rows in the first graph build of each view:", [int((ev[f"gap.own.{c}"] < 1e-4).sum()) for c in (0, n_calls)],
          "of", ev["gap.own.0"].numel(), "; over all builds", sum(int((ev[f"gap.own.{c}"] < 1e-4).sum()) for c in range(2 * n_calls)),
          "of", sum(ev[f"gap.own.{c}"].numel() for c in range(2 * n_calls)))
    del tape, ev, model

    enc, _ = deep_reference_encoder()
    model = SimCLR(CFG, enc)
    load_synth(model)
    fg = FixedGraph(model)
    model.eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss_eval = ntxent_loss(z_i, z_j, CFG)
    out.update(z_i_eval=z_i, z_j_eval=z_j, h_i_eval_clip=_per_clip(h_i), h_j_eval_clip=_per_clip(h_j), h_i_eval_head=h_i[:8],
               loss_eval=loss_eval.reshape(1).double())
    print("   eval (fixed graph) loss", float(loss_eval))

    # ---- step 0 on the fixed graph, view by view (graph builds 0 .. 23 belong to view i, 24 .. 47 to view j, in every pass)
    model.train()
    model.zero_grad()
    fg.call = 0
    with torch.no_grad():                           # pass 1: embeddings + running statistics (view i first, then view j)
        h_i, z_i = _view(model, x_i)
        h_j, z_j = _view(model, x_j)
    stats1 = _checksums((n, t.float().clone()) for n, t in model.state_dict().items() if n.endswith(("running_mean", "running_var")))
    keep = {n: t.clone() for n, t in model.state_dict().items()}
    zi, zj = z_i.clone().requires_grad_(True), z_j.clone().requires_grad_(True)
    loss = ntxent_loss(zi, zj, CFG)
    loss.backward()
    for first, x, dz, zref in ((0, x_i, zi.grad, z_i), (n_calls, x_j, zj.grad, z_j)):   # pass 2: the same forward with the tape
        fg.call = first
        _, z = _view(model, x)
        assert torch.equal(z.detach(), zref), "the re-forward of a view must reproduce pass 1 bit for bit"
        z.backward(dz)
        del z
    model.load_state_dict(keep)                     # the re-forwards updated the running statistics a second time
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0))
    out.update(z_i_train=z_i, z_j_train=z_j, h_i_train_clip=_per_clip(h_i), h_j_train_clip=_per_clip(h_j), h_i_train_head=h_i[:8],
               loss_train=np.array([float(loss.detach())], np.float64), gnorm=np.array([gn], np.float64))
    out.update({"grad." + n: grads[n] for n in (
        "encoder.stem.0.weight", "encoder.backbone.0.0.fc1.0.weight", "encoder.backbone.12.0.graph_conv.gconv.nn.0.weight",
        "encoder.backbone.26.1.fc2.1.weight", "encoder.proj.bias", "projector.2.bias")})
    save("deep_b256_k18", **out)
    with open(os.path.join(HERE, "deep_b256_k18_checksums.json"), "w") as f:
        json.dump({"grad": _checksums(grads.items()), "bn_after_step1": stats1}, f, indent=0)
    print("   step 0 (fixed graph): loss", float(loss.detach()), "gnorm", gn)


def _reference_function(path, name):
    """compile ONE function of a reference script that cannot be imported whole (faiss / torchaudio / dgl at module level)"""
    import ast
    with open(os.path.join(REF, path)) as f:
        tree = ast.parse(f.read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    return fn


def gold_fpdb():
    """Fingerprint-DB files written and read by the REFERENCE's own code. eval.py and test_fp.py import
    faiss / torchaudio / dgl at module level, so the two pieces are compiled from their syntax trees here, in the build
    container: `load_memmap_data` (eval.py:154-196) whole, and the writer statements of `create_ref_db`
    (test_fp.py:120-133: np.concatenate ... json.dump) as the body of a function of the variables they read.
    Only the resulting BYTES (data files) are committed."""
    import ast
    import shutil
    print("fpdb")
    ns = {"os": os, "np": np, "json": json}
    reader = _reference_function("eval.py", "load_memmap_data")
    exec(compile(ast.Module(body=[reader], type_ignores=[]), "eval.py", "exec"), ns)
    create = _reference_function("test_fp.py", "create_ref_db")
    first = next(i for i, st in enumerate(create.body) if not isinstance(st, ast.For) and "np.concatenate" in ast.unparse(st))
    tail = create.body[first:]                 # everything after the extraction loop: the statements of test_fp.py:120-133
    assert "np.memmap" in ast.unparse(tail[2]) and "json.dump" in ast.unparse(tail[-1]), [ast.unparse(t) for t in tail]
    writer = ast.FunctionDef(name="reference_writer",
                             args=ast.arguments(posonlyargs=[], args=[ast.arg(a) for a in
                                                                       ("fp", "z_i", "output_root_dir", "fname", "lookup_table")],
                                                kwonlyargs=[], kw_defaults=[], defaults=[]),
                             body=tail, decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[writer], type_ignores=[]))
    exec(compile(mod, "test_fp.py", "exec"), ns)
    out_dir = os.path.join(HERE, "fpdb")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    rng = np.random.default_rng(7)
    chunks = [rng.standard_normal((n, 128)).astype(np.float32) for n in (5, 1, 7)]       # per-song fingerprint blocks
    chunks = [c / np.linalg.norm(c, axis=1, keepdims=True) for c in chunks]
    chunks[2][3, 17] = np.nan                                                           # the reader zeroes NaNs in place
    names = ["songA"] * 5 + ["songB"] + ["songC"] * 7
    qnames = ["songA_0"] * 5 + ["songB_1"] + ["songC_2"] * 7
    for fname, lk in (("ref_db", names), ("query_db", qnames)):
        ns["reference_writer"](list(chunks), chunks[-1], out_dir, fname, lk)
    np.save(os.path.join(out_dir, "input_chunks.npy"), np.concatenate(chunks))
    np.save(os.path.join(out_dir, "input_chunk_sizes.npy"), np.array([5, 1, 7]))
    # what the reference's reader returns for these files (on a copy: it rewrites NaNs in the file it maps)
    tmp = os.path.join(out_dir, "_tmp")
    os.makedirs(tmp)
    for suffix in (".mm", "_shape.npy"):
        shutil.copy(os.path.join(out_dir, "ref_db" + suffix), os.path.join(tmp, "ref_db" + suffix))
    data, shape = ns["load_memmap_data"](tmp, "ref_db", display=False)
    np.save(os.path.join(out_dir, "reader_data.npy"), np.asarray(data))
    np.save(os.path.join(out_dir, "reader_shape.npy"), np.asarray(shape))
    data2, shape2 = ns["load_memmap_data"](tmp, "ref_db", append_extra_length=3, display=False)
    np.save(os.path.join(out_dir, "reader_extra3_shape.npy"), np.asarray(data2.shape))
    only_shape = ns["load_memmap_data"](tmp, "ref_db", shape_only=True)
    assert tuple(only_shape) == (13, 128)
    del data, data2
    shutil.rmtree(tmp)
    print("   files:", sorted(os.listdir(out_dir)))


def gold_shapes():
    """state_dict key names + shapes of SimCLR(GraphEncoder 't'): data for the state_dict-compat tests."""
    print("shapes")
    m = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t"))
    with open(os.path.join(HERE, "state_shapes.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in m.state_dict().items()}, f)


def gold_init():
    """per-key checksums of the reference's default initialisation under torch.manual_seed(42)"""
    print("init")
    torch.manual_seed(42)
    m = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t"))
    chk = {k: [float(v.double().sum()), float(v.double().norm())] for k, v in m.state_dict().items()
           if v.is_floating_point() and "relative_pos" not in k}
    with open(os.path.join(HERE, "init_seed42_checksums.json"), "w") as f:
        json.dump(chk, f, indent=0)


def gold_relpos():
    """the (dead) relative_pos buffers the reference's Grapher constructor builds, one per distinct (C, n)"""
    print("relpos")
    m = GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t")
    sd = m.state_dict()
    save("relative_pos_t", **{k: sd[k] for k in ("backbone.0.0.relative_pos", "backbone.3.0.relative_pos",
                                                 "backbone.6.0.relative_pos", "backbone.13.0.relative_pos")})


if __name__ == "__main__":
    torch.manual_seed(0)
    only = sys.argv[1:] or ["shapes", "init", "relpos", "knn", "mrconv", "block", "downsample", "peak", "ntxent", "e2e", "e2e_s", "e2e_m", "e2e_b",
                            "deep", "fpdb", "b256", "deep_b256"]
    for name in only:
        globals()["gold_" + name]()
