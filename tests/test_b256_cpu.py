"""CPU: the oracle at the TIMED size (B = 256, seed-42 default initialisation, the bench's clips) against the reference's
B = 256 goldens. The reference's own sensitivity at this size is part of the fixture (b256_seed42_k3_chaos.json): a 1e-7 relative
change of one view's input moves its loss by 2e-2 and its gradients by 30 %, so — like the B = 8 goldens — training mode is compared
with the reference's neighbour ids teacher-forced; eval mode (no cross-clip coupling) runs the oracle's own neighbour search."""
import torch

from b256_common import N_CALLS, bench_clips, chaos, check_tape, checksums, patches_of, per_clip
from oracle import ref_torch as R
from synth import GRAFP_CFG


def seed42_params():
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    torch.manual_seed(42)
    sd = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t")).state_dict()
    return {n: v.clone() for n, v in sd.items() if "relative_pos" not in n}


def test_reference_is_chaotic_in_training_mode_at_b256():
    """what the fixture says about the REFERENCE (measured by make_golden.py::gold_b256 on the live reference): summation order
    alone changes nothing, a 1e-7 input perturbation changes everything — the reason for teacher forcing at this size too"""
    c = chaos()
    quiet, loud = c["3 threads instead of 8 (summation order only)"], c["perturbation x_i*(1+1e-7)"]
    assert quiet["dloss"] < 1e-6 and quiet["flat_grad_rel"] < 1e-4 and min(quiet["rows_with_equal_ids_per_graph_build"]) == 1.0
    assert loud["dloss"] > 5e-3 and loud["flat_grad_rel"] > 0.1 and loud["gnorm_rel"] > 0.02
    rows = loud["rows_with_equal_ids_per_graph_build"]
    assert rows[0] > 0.999 and rows[11] < 0.9 and min(rows[12:]) == 1.0       # view i degrades block by block; view j untouched


def test_oracle_b256_eval_and_train_step0(golden):
    g = golden("b256_seed42_k3")
    chk = checksums()
    x_i, x_j = bench_clips()
    P = seed42_params()
    plan = R.encoder_plan("t", 3)
    torch.set_num_threads(8)
    with torch.no_grad():
        h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, False)
        loss_eval = float(R.ntxent(z_i, z_j, GRAFP_CFG["tau"]))
    # eval mode, own neighbour search: measured max |dz| 7.7e-5, |dloss| 1e-6 (80 k near-tie rows, none decisive)
    assert (z_i - g.t("z_i_eval")).abs().max() < 3e-4 and (z_j - g.t("z_j_eval")).abs().max() < 3e-4
    assert abs(loss_eval - float(g["loss_eval"][0])) < 1e-5
    hc = per_clip(h_i)
    assert ((hc[:, 1] - g.t("h_i_eval_clip")[:, 1]).abs() / g.t("h_i_eval_clip")[:, 1]).max() < 1e-4

    keys = R.trainable_keys(P)
    for k_ in keys:
        P[k_].requires_grad_(True)
    # the reference's graphs: own search, the reference's ids on its 1 121 near-tie rows; the per-clip hashes prove the rest
    R.TAPE = R.KnnTape(patch=patches_of(g))
    try:
        st = R.BNState()
        h_i, h_j, z_i, z_j = R.simclr_forward(x_i, x_j, P, GRAFP_CFG, plan, True, st)
        loss = R.ntxent(z_i, z_j, GRAFP_CFG["tau"])
        loss.backward()
        tape = R.TAPE
    finally:
        R.TAPE = None
    assert len(tape.recorded) == N_CALLS
    hard, soft, rows = check_tape(tape, g)
    print("kNN: hard", hard, "soft", soft, "of", rows)
    assert hard == 0 and soft <= 1500, (hard, soft)          # 1 121 recorded near-tie rows in the reference's own graphs
    # measured (teacher-forced): |dloss| 1.4e-6, gnorm 7.5e-5 relative, max |dz| 6.6e-6
    assert abs(float(loss.detach()) - float(g["loss_train"][0])) < 1e-5
    assert (z_i.detach() - g.t("z_i_train")).abs().max() < 5e-5 and (z_j.detach() - g.t("z_j_train")).abs().max() < 5e-5
    gn = float(torch.sqrt(sum(P[k_].grad.double().pow(2).sum() for k_ in keys if P[k_].grad is not None)))
    assert abs(gn - float(g["gnorm"][0])) / float(g["gnorm"][0]) < 5e-4
    # measured: late layers 2-3e-5; early layers 0.5-0.75 % — the fp32 noise floor of the reference's own early-layer gradients
    # (train-mode BatchNorm backward cancels heavily; at B = 8 it is 1.0-1.6 %)
    for name in [n for n in g if n.startswith("grad.")]:
        ref, got = g.t(name), P[name[5:]].grad
        if float(ref.norm()) < 1e-4:            # a conv bias in front of a train-mode BatchNorm: analytically zero
            continue
        rel = float((got - ref).norm() / ref.norm())
        late = name.startswith(("grad.encoder.backbone.14", "grad.encoder.proj", "grad.projector"))
        print(name, rel)
        assert rel < (1e-4 if late else 2e-2), (name, rel)
    for name, (s_, nrm) in chk["grad"].items():
        if nrm < 1e-4 or name not in P or P[name].grad is None:
            continue
        assert abs(float(P[name].grad.double().norm()) - nrm) <= 5e-3 * nrm, name
    for name, (s_, nrm) in chk["bn_after_step1"].items():
        assert abs(float(st.updates[name].double().norm()) - nrm) <= 1e-4 * max(nrm, 1.0), name


def test_bf16_emulation_fixture_matches_the_oracle_source():
    """tests/golden/b256_seed42_k3_bf16emu.* were computed by oracle/ref_torch.py with STORAGE = "bf16" (3 minutes of CPU) and record the
    digest of that source; the GPU test refuses a stale fixture — this is the same check where there is no GPU, so that an edit of the
    oracle is caught here first (regenerate with tests/golden/make_b256_emulation.py)"""
    import json
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    from make_b256_emulation import oracle_digest
    with open(os.path.join(gold, "b256_seed42_k3_bf16emu_checksums.json")) as f:
        assert json.load(f)["oracle_digest"] == oracle_digest()
