#!/usr/bin/env python3
"""Fixture of the bf16 EMULATION at the timed size (tests/test_b256_gpu.py): step 0 of train.py:53-75 evaluated by the oracle
(oracle/ref_torch.py, STORAGE = "bf16": a round-to-nearest-even bf16 rounding wherever the MI355X path stores an activation or stages a
GEMM operand in bf16), seed-42 default initialisation, the bench's 256 clip pairs, the REFERENCE's neighbour ids forced
(b256_seed42_k3.npz). ~3 minutes of CPU, so it is computed once here instead of in every GPU test run:

    python tests/golden/make_b256_emulation.py

The rounding points are this repo's statement of its own arithmetic — nothing in the reference pins them (it never runs reduced
precision, train.py:128); the fp32 goldens say what bf16 costs, this fixture says whether the HIP kernels implement exactly that
arithmetic. Regenerate it whenever oracle/ref_torch.py's STORAGE mode changes (tests/test_b256_cpu.py checks a digest of that file's
rounding helper against the one recorded here)."""
import hashlib
import inspect
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import ref_torch as R  # noqa: E402
from synth import GRAFP_CFG  # noqa: E402


def oracle_digest():
    """digest of the oracle source: the fixture is valid for exactly this restatement"""
    return hashlib.sha256(inspect.getsource(R).encode()).hexdigest()[:16]


def main():
    from b256_common import bench_clips, check_tape, patches_of, per_clip
    from conftest import Golden
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    with np.load(os.path.join(HERE, "b256_seed42_k3.npz")) as z:
        g = Golden({k: z[k] for k in z.files})
    torch.set_num_threads(8)
    torch.manual_seed(42)
    sd = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t")).state_dict()
    P = {n: v.clone() for n, v in sd.items() if "relative_pos" not in n}
    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    x_i, x_j = bench_clips()
    # the reference's step-0 graphs: one fp32 forward of the oracle, its own search + the fixture's near-tie rows, proven by the hashes
    R.TAPE = tape = R.KnnTape(patch=patches_of(g))
    try:
        with torch.no_grad():
            R.simclr_forward(x_i, x_j, P, GRAFP_CFG, R.encoder_plan("t", 3), True, R.BNState())
    finally:
        R.TAPE = None
    hard, soft, rows = check_tape(tape, g)
    assert hard == 0, (hard, soft, rows)
    R.STORAGE = "bf16"
    R.TAPE = R.KnnTape(replay=tape.patched)
    try:
        st = R.BNState()
        h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, R.encoder_plan("t", 3), True, st)
        loss = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
        loss.backward()
    finally:
        R.STORAGE = None
        R.TAPE = None
    grads = {k_: P[k_].grad for k_ in keys if P[k_].grad is not None}
    gn = float(torch.sqrt(sum(v.double().pow(2).sum() for v in grads.values())))
    out = {"z_i": z_i.detach().numpy(), "z_j": z_j.detach().numpy(), "h_i_clip": per_clip(h_i).numpy(),
           "h_j_clip": per_clip(h_j).numpy(), "h_i_head": h_i[:8].detach().numpy(),
           "loss": np.array([float(loss.detach())]), "gnorm": np.array([gn])}
    for name in [n for n in g if n.startswith("grad.")]:
        out[name] = grads[name[5:]].numpy()
    np.savez_compressed(os.path.join(HERE, "b256_seed42_k3_bf16emu.npz"), **out)
    chk = {"oracle_digest": oracle_digest(),
           "grad": {n: [float(v.double().sum()), float(v.double().norm())] for n, v in grads.items()},
           "bn_after_step1": {n: [float(v.double().sum()), float(v.double().norm())] for n, v in st.updates.items()
                              if n.endswith(("running_mean", "running_var"))}}
    with open(os.path.join(HERE, "b256_seed42_k3_bf16emu_checksums.json"), "w") as f:
        json.dump(chk, f, indent=0)
    print("bf16 emulation at B = 256: loss", float(loss.detach()), "(fp32 reference", float(g["loss_train"][0]), ") gnorm", gn,
          "(", float(g["gnorm"][0]), ") max |dz|", float((z_i.detach() - g.t("z_i_train")).abs().max()))


if __name__ == "__main__":
    main()
