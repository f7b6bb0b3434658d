"""GPU: the HIP path at the TIMED size — B = 256 clip pairs (the bench's own inputs), seed-42 default initialisation, k = 3 —
against goldens produced by the live reference at that size (tests/golden/make_golden.py::gold_b256).

Every other reference-pinned end-to-end test runs B = 8 (B = 4 for the deep plan); the kernel instantiations bench.py times are
chosen by SHAPE (full-tile / 64-deep-stage GEMMs, 128x64 and 8-wave weight-gradient tiles, split-major grids, the capped
BatchNorm-backward apply pass, the two-way channel split of the aggregation), so they are only reached at this size. The launch
counters of the library (nsid_debug_counter) prove that those variants are what ran.

Training mode is chaotic at this size too — measured ON THE REFERENCE and committed with the fixture
(b256_seed42_k3_chaos.json: a 1e-7 relative change of view i's input moves the reference's own loss by 1.9e-2, its gradient norm by
6.5 % and its gradients by 34 %) — so step 0 is compared with the reference's neighbour ids teacher-forced, the neighbour SETS the
HIP search produces on the same features are compared with the reference's outside recorded near-ties, and the free-running step
(own neighbour search, nothing forced) is bounded by the reference's own response to that perturbation."""
import json
import os

import pytest
import torch

from b256_common import B, N_CALLS, bench_clips, chaos, check_tape, checksums, patches_of, per_clip
from conftest import ROOT
from compare import maxerr, relerr
from synth import GRAFP_CFG

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEASURED = {}


def note(key, value):
    MEASURED[key] = value
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "b256_measured.json"), "w") as f:
            json.dump(MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass


def build(overlap=False):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    torch.manual_seed(42)
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t"), overlap_views=overlap)


def set_mode(mode):
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    ops.set_gemm_precision(mode)
    F_.set_activation_dtype(mode)


@pytest.fixture()
def restore_mode():
    yield
    from neuralsampleid_amd import functional as F_
    F_.TAPE = None
    set_mode("fp32")


def hip_step0(x_i, x_j, tape, overlap=False, patch=None):
    """step 0 of train.py:53-75 with FusedClipAdam(direct_grads) — the optimiser the bench uses; tape: neighbour ids to force;
    patch: instead, the reference's ids on its near-tie rows only (b256_common.patches_of) — the returned "knn_tape" then holds the
    own search's graphs (.recorded) and the ones used (.patched) for b256_common.check_tape"""
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.ntxent import ntxent_loss
    model = build(overlap).to(DEV).train()
    opt = FusedClipAdam(model.parameters(), lr=GRAFP_CFG["lr"], max_norm=1.0)
    F_.TAPE = knn_tape = F_.KnnTape(replay=tape, patch=patch)
    try:
        opt.zero_grad()
        h_i, h_j, z_i, z_j = model(x_i, x_j)
        loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
        loss.backward()
        rec = [t.clone() for t in F_.TAPE.recorded]
    finally:
        F_.TAPE = None
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None and p.requires_grad}
    opt.step()
    torch.cuda.synchronize()
    return dict(model=model, h_i=h_i.detach(), h_j=h_j.detach(), z_i=z_i.detach(), z_j=z_j.detach(), loss=float(loss.detach()),
                gnorm=float(opt.grad_norm), grads=grads, tape=rec, knn_tape=knn_tape)


_REFERENCE_TAPE = []


def reference_tape(g, x_i, x_j):
    """the reference's 24 step-0 graphs, rebuilt once per session in strict fp32: own search + the fixture's near-tie rows, every clip
    of every build proven by its hash (b256_common.check_tape). Leaves the arithmetic mode at fp32."""
    if not _REFERENCE_TAPE:
        from neuralsampleid_amd import functional as F_
        set_mode("fp32")
        model = build().to(DEV).train()
        F_.TAPE = tape = F_.KnnTape(patch=patches_of(g))
        try:
            with torch.no_grad():
                model(x_i, x_j)
        finally:
            F_.TAPE = None
        hard, soft, rows = check_tape(tape, g)
        assert hard == 0, (hard, soft, rows)
        _REFERENCE_TAPE.extend(t.clone() for t in tape.patched)
    return list(_REFERENCE_TAPE)


LATE = ("encoder.backbone.14", "encoder.proj", "projector")


def grad_report(grads, g, chk):
    """relative L2 of the fully stored gradients + the worst relative difference of per-parameter gradient NORMS (all 300 keys)"""
    full = {}
    for name in [n for n in g if n.startswith("grad.")]:
        ref = g.t(name)
        if float(ref.norm()) < 1e-4:
            continue
        full[name[5:]] = relerr(grads[name[5:]], ref)
    norms = {}
    for name, (s_, nrm) in chk["grad"].items():
        if nrm < 1e-4 or name not in grads:
            continue
        norms[name] = abs(float(grads[name].double().norm()) - nrm) / nrm
    return full, norms


def test_fp32_hip_vs_reference_at_b256(golden, restore_mode):
    """strict-fp32 HIP path against the reference at the timed size: eval forward (own neighbour search), step 0 with the
    reference's neighbour ids forced (loss, embeddings, global and per-parameter gradients, running statistics, neighbour sets),
    and the free-running step 0 bounded by the reference's own chaos"""
    g = golden("b256_seed42_k3")
    chk = checksums()
    set_mode("fp32")
    x_i, x_j = (t.to(DEV) for t in bench_clips())
    model = build().to(DEV).eval()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(x_i, x_j)
    m = {"eval_max_dz": max(maxerr(z_i, g.t("z_i_eval")), maxerr(z_j, g.t("z_j_eval"))),
         "eval_h_clip_norm_rel": float(((per_clip(h_i)[:, 1] - g.t("h_i_eval_clip")[:, 1]).abs() / g.t("h_i_eval_clip")[:, 1]).max()),
         "eval_h_head_max": maxerr(h_i[:8], g.t("h_i_eval_head"))}
    del model
    r = hip_step0(x_i, x_j, None, patch=patches_of(g))
    hard, soft, rows = check_tape(r["knn_tape"], g)
    ref_tape = [t.clone() for t in r["knn_tape"].patched]              # hard == 0 (asserted below): these ARE the reference's graphs
    full, norms = grad_report(r["grads"], g, chk)
    sd = r["model"].state_dict()
    bn = max(abs(float(sd[n].double().norm()) - nrm) / max(nrm, 1.0) for n, (s_, nrm) in chk["bn_after_step1"].items())
    m.update({"dloss": abs(r["loss"] - float(g["loss_train"][0])), "gnorm_rel": abs(r["gnorm"] - float(g["gnorm"][0])) / float(g["gnorm"][0]),
              "max_dz": max(maxerr(r["z_i"], g.t("z_i_train")), maxerr(r["z_j"], g.t("z_j_train"))),
              "h_clip_norm_rel": float(((per_clip(r["h_i"])[:, 1] - g.t("h_i_train_clip")[:, 1]).abs() / g.t("h_i_train_clip")[:, 1]).max()),
              "knn_hard": hard, "knn_soft": soft, "knn_rows": rows, "full_grads": full,
              "grad_norm_rel_worst": max(norms.values()), "grad_norm_rel_worst_name": max(norms, key=norms.get),
              "grad_norm_rel_late": max(v for n, v in norms.items() if n.startswith(LATE)), "bn_norm_rel": bn})
    # free-running: own neighbour search in every block, nothing forced
    f = hip_step0(x_i, x_j, None)
    same = [float((torch.sort(a.long(), -1).values == torch.sort(b.to(DEV).long(), -1).values).all(-1).float().mean())
            for a, b in zip(f["tape"], ref_tape)]
    m["free"] = {"dloss": abs(f["loss"] - float(g["loss_train"][0])), "gnorm_rel": abs(f["gnorm"] - float(g["gnorm"][0])) / float(g["gnorm"][0]),
                 "max_dz": max(maxerr(f["z_i"], g.t("z_i_train")), maxerr(f["z_j"], g.t("z_j_train"))),
                 "rows_with_equal_sets_per_graph_build": same}
    note("fp32_vs_reference", m)
    print("measured", json.dumps(m, indent=1))
    # eval: no cross-clip coupling; the oracle itself (CPU, own search) sits at max |dz| 7.7e-5 from these goldens
    assert m["eval_max_dz"] < TOL32["eval_max_dz"] and m["eval_h_clip_norm_rel"] < TOL32["eval_h_clip_norm_rel"]
    # forced step 0
    assert m["dloss"] < TOL32["dloss"] and m["gnorm_rel"] < TOL32["gnorm_rel"] and m["max_dz"] < TOL32["max_dz"]
    assert m["h_clip_norm_rel"] < TOL32["h_clip_norm_rel"] and m["bn_norm_rel"] < TOL32["bn_norm_rel"]
    assert hard == 0 and soft <= TOL32["knn_soft"], (hard, soft)
    for n, e in full.items():
        assert e < (TOL32["grad_late"] if n.startswith(LATE) else TOL32["grad_early"]), (n, e)
    assert m["grad_norm_rel_late"] < TOL32["grad_norm_late"] and m["grad_norm_rel_worst"] < TOL32["grad_norm_worst"], m["grad_norm_rel_worst_name"]
    # free-running step: within 3x of what a 1e-7 input perturbation does to the reference itself
    ref_resp = chaos()["perturbation x_i*(1+1e-7)"]
    assert m["free"]["dloss"] < 3 * ref_resp["dloss"] and m["free"]["gnorm_rel"] < 3 * ref_resp["gnorm_rel"]
    assert m["free"]["max_dz"] < 3 * ref_resp["max_dz"]
    assert same[0] > 0.999 and same[12] > 0.999                       # the first graph of each view sees identical features


# bounds: <= 3x the values measured on MI355X in round 3 (gpurun_out/c1/b256_measured.json) — all far inside VERDICT r2's targets
# (loss 1e-4, global norm 1 %). Measured: eval max |dz| 8.0e-5 (the CPU oracle with its own search: 7.7e-5), per-clip |h| 3.6e-5;
# forced step 0: |dloss| 9.5e-7 (two float32 ulps of 4.64), global gradient norm 3.5e-4, max |dz| 5.4e-6, per-clip |h| 2.6e-6,
# running statistics 3e-8, neighbour sets: 0 hard mismatches, 10 rows on recorded near-ties of 622 592; fully stored gradients:
# late layers 1.7-2.6e-5, early layers 4.3-6.5e-3 (the reference's own fp32 floor there: the CPU oracle sits at 5.2-7.5e-3);
# per-parameter gradient norms: late 2.2e-4, worst 1.9e-3. Free-running step (nothing forced): |dloss| 9.0e-3, norm 0.7 %,
# max |dz| 0.27, rows with equal neighbour sets falling from 1.0 to 0.73 over the 12 blocks of a view — the same decay the
# reference shows under a 1e-7 input perturbation (0.67).
TOL32 = {"eval_max_dz": 2.5e-4, "eval_h_clip_norm_rel": 1.2e-4,
         "dloss": 5e-6, "gnorm_rel": 1.1e-3, "max_dz": 2e-5, "h_clip_norm_rel": 1e-5, "bn_norm_rel": 1e-6, "knn_soft": 40,
         "grad_late": 1e-4, "grad_early": 2e-2, "grad_norm_late": 7e-4, "grad_norm_worst": 6e-3}


def emulation_fixture(golden):
    """step 0 by the ORACLE with STORAGE = "bf16" at this size, precomputed by tests/golden/make_b256_emulation.py (3 minutes of
    CPU): a bf16 rounding wherever the MI355X path stores an activation or stages a GEMM operand in bf16 — builder-authored rounding
    points: nothing in the reference pins them. The fixture records the digest of the oracle source it was
    computed with; a changed oracle must regenerate it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_b256_emulation import oracle_digest
    with open(os.path.join(ROOT, "tests", "golden", "b256_seed42_k3_bf16emu_checksums.json")) as f:
        chk = json.load(f)
    assert chk["oracle_digest"] == oracle_digest(), "oracle/ref_torch.py changed: rerun tests/golden/make_b256_emulation.py"
    return golden("b256_seed42_k3_bf16emu"), chk


def test_bf16_hip_vs_emulation_at_b256_and_timed_variants_ran(golden, restore_mode):
    """the TIMED arithmetic (bf16 storage, bf16 MFMA operands, fp32 accumulate) at the timed size, reference neighbour ids forced:
    (a) against the oracle's bf16 emulation at the same size, (b) against the reference's fp32 goldens next to the emulation's own
    distance from them, (c) the launch counters show that the kernel variants bench.py times are the ones that just ran"""
    from neuralsampleid_amd._lib import launch_counters
    g = golden("b256_seed42_k3")
    chk = checksums()
    em, em_chk = emulation_fixture(golden)
    x_i, x_j = (t.to(DEV) for t in bench_clips())
    ref_tape = reference_tape(g, x_i, x_j)
    set_mode("bf16")
    launch_counters(reset=True)
    r = hip_step0(x_i, x_j, ref_tape)
    cnt = launch_counters()
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.detach().cpu().float(), b.float(), dim=1).min())
    # fully stored gradients: relative L2; all parameters: relative difference of the gradient NORMS
    rel_g = {n[5:]: relerr(r["grads"][n[5:]], em.t(n)) for n in em if n.startswith("grad.") and float(em.t(n).norm()) > 1e-4}
    nrm_g = {n: abs(float(r["grads"][n].double().norm()) - nrm) / nrm for n, (s_, nrm) in em_chk["grad"].items()
             if nrm > 1e-4 and n in r["grads"]}
    hc = per_clip(r["h_i"])
    a = {"h_clip_norm_rel": float(((hc[:, 1] - em.t("h_i_clip")[:, 1]).abs() / em.t("h_i_clip")[:, 1]).max()),
         "rel_h_head": relerr(r["h_i"][:8], em.t("h_i_head")),
         "max_dz": max(maxerr(r["z_i"], em.t("z_i")), maxerr(r["z_j"], em.t("z_j"))),
         "cos_z_min": min(cos(r["z_i"], em.t("z_i")), cos(r["z_j"], em.t("z_j"))), "dloss": abs(r["loss"] - float(em["loss"][0])),
         "gnorm_rel": abs(r["gnorm"] - float(em["gnorm"][0])) / float(em["gnorm"][0]),
         "grad_rel": rel_g, "grad_rel_late": max(v for n, v in rel_g.items() if n.startswith(LATE)),
         "grad_rel_early": max(v for n, v in rel_g.items() if not n.startswith(LATE)),
         "grad_norm_rel_worst": max(nrm_g.values()), "grad_norm_rel_worst_name": max(nrm_g, key=nrm_g.get),
         "grad_norm_rel_median": sorted(nrm_g.values())[len(nrm_g) // 2]}
    sd = r["model"].state_dict()
    a["bn_norm_rel"] = max(abs(float(sd[n].double().norm()) - nrm) / max(nrm, 1.0) for n, (s_, nrm) in em_chk["bn_after_step1"].items())

    def dev(z_i, loss, gnorm, grads):
        full, norms = grad_report(grads, g, chk)
        return {"cos_z_min": cos(z_i, g.t("z_i_train")), "dloss": abs(loss - float(g["loss_train"][0])),
                "gnorm_rel": abs(gnorm - float(g["gnorm"][0])) / float(g["gnorm"][0]), "full_grads": full}
    b_hip = dev(r["z_i"], r["loss"], r["gnorm"], r["grads"])
    b_em = dev(em.t("z_i"), float(em["loss"][0]), float(em["gnorm"][0]), {n[5:]: em.t(n) for n in em if n.startswith("grad.")})
    em_norms = {n: abs(nrm - chk["grad"][n][1]) / chk["grad"][n][1] for n, (s_, nrm) in em_chk["grad"].items()
                if n in chk["grad"] and chk["grad"][n][1] > 1e-4}
    b_em["grad_norm_rel_median"] = sorted(em_norms.values())[len(em_norms) // 2]
    note("bf16_vs_emulation", {"vs_emulation": a, "hip_vs_golden": b_hip, "emulation_vs_golden": b_em, "counters": cnt})
    print("measured", json.dumps({"vs_emulation": a, "hip_vs_golden": b_hip, "emulation_vs_golden": b_em, "counters": cnt}, indent=1))
    # (c) the variants of the timed step
    for key in ("gemm_full", "gemm_ks2", "gemm_split_major", "gemm_affine_load", "gemm_bn_sums", "wgrad_grouped", "wgrad_grouped_w3",
                "bn_bwd_apply_capped", "knn2", "mr_fwd_lds"):       # (the weight gradients: the deferred grouped phase, functional.DEFER_WGRAD)
        assert cnt[key] > 0, (key, cnt)
    assert cnt["gemm256"] == 0 and cnt["knn_strips"] == 0, cnt     # default tuning: not in the training step
    # (a) same rounding points on both sides
    assert a["h_clip_norm_rel"] < TOL16["h_clip_norm_rel"] and a["max_dz"] < TOL16["max_dz"] and a["cos_z_min"] > TOL16["cos_z_min"]
    assert a["dloss"] < TOL16["dloss"] and a["gnorm_rel"] < TOL16["gnorm_rel"] and a["bn_norm_rel"] < TOL16["bn_norm_rel"]
    assert a["grad_rel_late"] < TOL16["grad_rel_late"] and a["grad_rel_early"] < TOL16["grad_rel_early"]
    assert a["grad_norm_rel_median"] < TOL16["grad_norm_rel_median"]
    assert a["grad_norm_rel_worst"] < TOL16["grad_norm_rel_worst"], a["grad_norm_rel_worst_name"]
    # (b) no further from the reference than the emulated arithmetic is
    assert b_hip["dloss"] < 1.5 * b_em["dloss"] + 5e-3 and b_hip["gnorm_rel"] < 1.5 * b_em["gnorm_rel"] + 5e-3
    assert b_hip["cos_z_min"] > 1.0 - 1.5 * (1.0 - b_em["cos_z_min"]) - 1e-3
    for n, e in b_hip["full_grads"].items():
        assert e < 1.3 * b_em["full_grads"][n] + 0.05, (n, e, b_em["full_grads"][n])


# <= 3x the values measured on MI355X in round 3 (gpurun_out/c1/b256_measured.json). HIP bf16 against the emulation: per-clip |h|
# 1.5 %, max |dz| 0.026, min cos z 0.9969, |dloss| 8.1e-3, running statistics 5e-5, per-parameter gradient NORMS: median 0.45 %,
# worst 25 % (the peak extractor's bias); fully stored gradients: late layers 6-9 %, early layers 29-36 % relative L2; global norm
# 9.4 %. The global norm is NOT the well-conditioned quantity it is at batch 8: with the default initialisation 116 of its 157 come
# from the peak extractor's conv weight, the first layer of the net and the most sensitive gradient there is (the fp32 reference
# itself moves it by 40 % under a 1e-7 input perturbation). Against the fp32 goldens both sit equally far: |dloss| 0.046 (HIP) /
# 0.038 (emulation), early gradients 0.60-0.69 / 0.64-0.69, late 0.19-0.26 / 0.18-0.26, min cos z 0.977 / 0.978 — bf16 storage of
# 60 activation tensors per view costs that much in a train-mode step; the loss curve of 30 steps is what test_training_curve
# judges it on.
# Round 4 (VERDICT r3 "what's weak" 1): a relative-L2 bound >= 1 cannot fail (a ZERO gradient scores 1.0), so the three bounds that sat
# there at 3x measured are cut to 1.5x: early-layer gradients 0.36 measured -> 0.55, the worst per-parameter norm 0.24 -> 0.40, the
# global norm 0.09 -> 0.15. A summation-order change in a kernel can move these chaotic quantities; it then has to be re-measured and
# argued, which is the point.
TOL16 = {"h_clip_norm_rel": 0.045, "max_dz": 0.08, "cos_z_min": 0.9908, "dloss": 0.025, "gnorm_rel": 0.15, "bn_norm_rel": 1.6e-4,
         "grad_rel_late": 0.28, "grad_rel_early": 0.55, "grad_norm_rel_median": 0.014, "grad_norm_rel_worst": 0.40}
