#!/usr/bin/env python3
"""diagnosis: per-parameter gradient differences of one B = 8 training step between the tile GEMMs (ws_gemm = 0) and the
weight-stationary forms (ws_gemm = 1 / 2 / 3), reference kNN ids forced"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from conftest import load_golden
from synth import GRAFP_CFG, synth_state
from neuralsampleid_amd import functional as F_, ops
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.optim import FusedClipAdam
from neuralsampleid_amd.simclr.ntxent import ntxent_loss
from neuralsampleid_amd.simclr.simclr import SimCLR

k = 3
g = load_golden(f"e2e_b8_k{k}")
n = sum(1 for key in g if key.startswith("knn.s0."))
tape = [g.t(f"knn.s0.{c}") for c in range(n)]
ops.set_gemm_precision("bf16"); F_.set_activation_dtype("bf16")
res = {}
for mode in (0, 1, 2, 3, 0):
    ops.set_tuning("ws_gemm", mode)
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=k, size="t"))
    model.load_state_dict(synth_state(model.state_dict()))
    model.to("cuda").train()
    opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
    F_.TAPE = F_.KnnTape(replay=tape)
    opt.zero_grad()
    ops.launch_counters(reset=True)
    h_i, h_j, z_i, z_j = model(g.t("x_i").cuda(), g.t("x_j").cuda())
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    F_.TAPE = None
    c = ops.launch_counters()
    grads = {nm: p.grad.detach().float().cpu().clone() for nm, p in model.named_parameters() if p.grad is not None}
    opt.step(); torch.cuda.synchronize()
    res.setdefault(mode, []).append((float(loss), float(opt.grad_norm), grads, h_i.detach().float().cpu()))
    print(f"mode {mode}: loss {float(loss):.6f} gnorm {float(opt.grad_norm):.5f} ws_fwd {c['ws_fwd']} ws_bwd {c['ws_bwd_data']}", flush=True)
base = res[0][0]
print("repeat of mode 0: gnorm", res[0][1][1], "max rel grad diff", max(float((res[0][1][2][n_] - base[2][n_]).norm() / (base[2][n_].norm() + 1e-12)) for n_ in base[2]))
for mode in (1, 2, 3):
    r = res[mode][0]
    rel = {n_: float((r[2][n_] - base[2][n_]).norm() / (base[2][n_].norm() + 1e-12)) for n_ in base[2]}
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:8]
    print(f"mode {mode}: dh {float((r[3] - base[3]).norm() / base[3].norm()):.4f} gnorm {r[1]:.5f} vs {base[1]:.5f}; median rel {sorted(rel.values())[len(rel)//2]:.4f}; worst {worst}")
    nr = {n_: float(r[2][n_].norm() / (base[2][n_].norm() + 1e-12)) for n_ in base[2]}
    odd = sorted(nr.items(), key=lambda kv: -abs(kv[1] - 1))[:8]
    print("   norm ratios furthest from 1:", [(a, round(b, 3)) for a, b in odd])
