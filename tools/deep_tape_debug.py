#!/usr/bin/env python3
"""Which kNN call / row of the deep configuration's train-mode forward disagrees with the reference's neighbour sets
(the body of tests/test_e2e_gpu.py::test_deep_config4_e2e, with per-call reporting)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, test_e2e_gpu as T
from neuralsampleid_amd import functional as F_
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
g = conftest.load_golden("deep_b4_k18")
model = SimCLR(T.GRAFP_CFG, GraphEncoder(T.GRAFP_CFG, in_channels=T.GRAFP_CFG["n_filters"], k=18, size="t",
                                         blocks=[4, 4, 12, 4], use_dilation=True))
T.load_synth(model)
x_i, x_j = g.t("x_i").to(T.DEV), g.t("x_j").to(T.DEV)
if len(sys.argv) > 1:       # the call of test_knn_module_layout first (the failure depends on what ran before)
    from neuralsampleid_amd.encoder.gcn_lib.torch_edge import DenseDilatedKnnGraph
    xx = conftest.load_golden("knn_c64n256").t("x").to(T.DEV)
    ei = DenseDilatedKnnGraph(4, 2)(xx)
    print("pre-call done", ei.shape)
for tag, train in (("eval", False), ("s0", True)):
    model.train(train)
    gold_idx, gaps = T.tape_of(g, tag)
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    with torch.set_grad_enabled(train):
        model(x_i, x_j)
    rec = F_.TAPE.recorded
    F_.TAPE = None
    for c, (r, gi, gp) in enumerate(zip(rec, gold_idx, gaps)):
        a = np.sort(r.cpu().numpy(), axis=-1); b = np.sort(np.asarray(gi), axis=-1)
        bad = (a != b).any(-1); hard = bad & (np.asarray(gp) >= 1e-4)
        if hard.any():
            for pos in np.argwhere(hard):
                pos = tuple(pos)
                print(tag, "call", c, "shape", a.shape, "row", pos, "gap", float(np.asarray(gp)[pos]))
                print("   got ", a[pos]); print("   gold", b[pos])
                print("   raw ", r.cpu().numpy()[pos])
    print(tag, "done")
