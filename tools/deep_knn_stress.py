#!/usr/bin/env python3
"""Is the deep configuration's kNN tape reproducible? Repeats the teacher-forced train-mode forward of
tests/test_e2e_gpu.py::test_deep_config4_e2e with the allocator's free blocks and every CU's LDS filled with garbage in between,
and reports any call whose recorded neighbour ids change (or disagree with the reference outside near-ties)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, test_e2e_gpu as T
from neuralsampleid_amd import functional as F_, ops
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
g = conftest.load_golden("deep_b4_k18")
model = SimCLR(T.GRAFP_CFG, GraphEncoder(T.GRAFP_CFG, in_channels=T.GRAFP_CFG["n_filters"], k=18, size="t",
                                         blocks=[4, 4, 12, 4], use_dilation=True))
T.load_synth(model)
x_i, x_j = g.t("x_i").to(T.DEV), g.t("x_j").to(T.DEV)
model.train(True)
gold_idx, gaps = T.tape_of(g, "s0")
first = None
gen = torch.Generator(device="cuda").manual_seed(1)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    junk = torch.randint(-2**31, 2**31 - 1, (32 << 20,), device="cuda", dtype=torch.int32, generator=gen)     # 128 MB of garbage
    del junk
    rr = torch.randn(1024 * 256, 64, device="cuda", generator=gen)
    ops.knn_graph(rr, 1024, 256, 64, 4, 2, None)              # knn2 on every CU: leaves its LDS image behind
    del rr
    F_.TAPE = F_.KnnTape(replay=gold_idx)
    model(x_i, x_j)
    rec = [r.cpu().numpy() for r in F_.TAPE.recorded]
    F_.TAPE = None
    for c, (r, gi, gp) in enumerate(zip(rec, gold_idx, gaps)):
        a = np.sort(r, axis=-1); b = np.sort(np.asarray(gi), axis=-1)
        hard = (a != b).any(-1) & (np.asarray(gp) >= 1e-4)
        for pos in np.argwhere(hard)[:3]:
            pos = tuple(pos)
            print(f"it {it} call {c} {r.shape} row {pos} gap {float(np.asarray(gp)[pos]):.3g}\n   got  {r[pos]}\n   gold {np.asarray(gi)[pos]}")
        if first is not None and (first[c] != r).any():
            bad = np.argwhere((first[c] != r).any(-1))
            print(f"it {it} call {c} {r.shape}: {len(bad)} rows differ from iteration 0, e.g. {tuple(bad[0])}\n   now   {r[tuple(bad[0])]}\n   first {first[c][tuple(bad[0])]}")
    if first is None:
        first = rec
print("done")
