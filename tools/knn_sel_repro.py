#!/usr/bin/env python3
"""Reproducibility of knn_sel_kernel on the deep configuration's own features: captures the arguments of every N = 256 kNN call of
one teacher-forced train-mode forward (tests/golden/deep_b4_k18), then repeats each call and compares the neighbour ids (and, with
a -DNSID_KSEL_DUMP build, the distance keys) with the first run.   Usage: [NSID_LIB=...] python tools/knn_sel_repro.py [repeats]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, test_e2e_gpu as T
from neuralsampleid_amd import functional as F_, ops
from neuralsampleid_amd._lib import lib
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = conftest.load_golden("deep_b4_k18")
model = SimCLR(T.GRAFP_CFG, GraphEncoder(T.GRAFP_CFG, in_channels=T.GRAFP_CFG["n_filters"], k=18, size="t",
                                         blocks=[4, 4, 12, 4], use_dilation=True))
T.load_synth(model)
x_i, x_j = g.t("x_i").to(T.DEV), g.t("x_j").to(T.DEV)
model.train(True)
gold_idx, gaps = T.tape_of(g, "s0")
calls = []
orig = ops.knn_graph
def capture(r, B, N, C, k, dilation=1, aff=None):
    if N == 256:
        calls.append((r.clone(), B, N, C, k, dilation, None if aff is None else ops.BNAffine(aff.scale.clone(), aff.shift.clone())))
    return orig(r, B, N, C, k, dilation, aff)
ops.knn_graph = capture
F_.TAPE = F_.KnnTape(replay=gold_idx)
model(x_i, x_j)
F_.TAPE = None
ops.knn_graph = orig
print("captured", len(calls), "calls with N = 256")
for ci, (r, B, N, C, k, d, aff) in enumerate(calls):
    keys = torch.zeros(B * 256 * 256, dtype=torch.int32, device="cuda")
    assert lib.nsid_debug_knn_trace(keys.data_ptr()) == 0
    first = orig(r, B, N, C, k, d, aff).cpu().numpy()
    k0 = keys.cpu().numpy().copy()
    have_keys = bool(k0.any())
    bad = 0
    for it in range(reps):
        keys.zero_()
        now = orig(r, B, N, C, k, d, aff).cpu().numpy()
        if (now != first).any():
            bad += 1
            if bad <= 2:
                rows = np.argwhere((now != first).any(-1))
                print(f"call {ci} repeat {it}: rows {[tuple(int(v) for v in x) for x in rows[:4]]} differ")
                for rw in rows[:1]:
                    rw = tuple(rw)
                    print("   now  ", now[rw]); print("   first", first[rw])
                    if have_keys:
                        k1 = keys.cpu().numpy()
                        dk = np.argwhere(k1 != k0)
                        print("   keys that differ:", len(dk), "e.g. flat", dk[:12, 0].tolist(),
                              "-> (clip,row,col)", [(int(i) >> 16, (int(i) >> 8) & 255, int(i) & 255) for i in dk[:12, 0]])
                        def val(kk):          # key -> distance
                            kk = kk.astype(np.uint32)
                            bits = np.where(kk >> 31, kk ^ np.uint32(0x80000000), ~kk)
                            return bits.view(np.float32)
                        ii = dk[:16, 0]
                        print("   now  ", np.round(val(k1[ii]), 5).tolist())
                        print("   first", np.round(val(k0[ii]), 5).tolist())
    lib.nsid_debug_knn_trace(None)
    print(f"call {ci}: {bad} of {reps} repeats differ from the first run (keys dumped: {have_keys})")
