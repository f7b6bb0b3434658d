"""diagnosis: per-parameter gradient difference between an in-chain and a deferred step (same forced neighbour ids)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import torch
from neuralsampleid_amd import functional as F_, ops
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.optim import FusedClipAdam
from neuralsampleid_amd.simclr.ntxent import ntxent_loss
from neuralsampleid_amd.simclr.simclr import SimCLR
from synth import GRAFP_CFG, synth_clips, synth_state
ops.set_gemm_precision("bf16"); F_.set_activation_dtype("bf16")
x_i, x_j = synth_clips(8)
res = {}
tape = None
for tag, defer, grouped in (("base", 0, 1), ("base2", 0, 1), ("defer_items", 1, 0), ("defer", 1, 1)):
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=8, k=3, size="t"))
    model.load_state_dict(synth_state(model.state_dict())); model.cuda().train()
    opt = FusedClipAdam(model.parameters(), lr=1e-4, max_norm=1.0)
    F_.DEFER_WGRAD = defer
    ops.WGRAD_GROUPED = grouped
    F_.TAPE = F_.KnnTape(replay=tape)
    opt.zero_grad()
    _, _, z_i, z_j = model(x_i.cuda(), x_j.cuda())
    loss = ntxent_loss(z_i, z_j, GRAFP_CFG)
    loss.backward()
    if tape is None:
        tape = [t.clone() for t in F_.TAPE.recorded]
    F_.TAPE = None
    torch.cuda.synchronize()
    res[tag] = (z_i.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad})
for tag in ("base2", "defer_items", "defer"):
    print(tag, "dz", float((res[tag][0] - res["base"][0]).abs().max()), "loss", res[tag][1], res["base"][1])
    worst = []
    for n, g in res[tag][2].items():
        ref = res["base"][2][n]
        worst.append((float((g - ref).norm()) / max(float(ref.norm()), 1e-9), n, float(ref.norm())))
    worst.sort(reverse=True)
    print("  differing params:", sum(1 for w in worst if w[0] > 1e-5), "of", len(worst))
    for w in worst[:25]:
        print("   ", w)
    print("    stem:", [w for w in worst if "stem.0.weight" in w[1]])
