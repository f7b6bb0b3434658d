"""diagnosis: which recorded weight-gradient operands change between record time and flush time"""
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
model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=8, k=3, size="t"))
model.load_state_dict(synth_state(model.state_dict())); model.cuda().train()
opt = FusedClipAdam(model.parameters(), lr=1e-4, max_norm=1.0)
snaps = []
orig_add = F_.DeferredWgrads.add
def add(self, dout, x, dw, *rest):
    snaps.append((dout, dout.clone(), x, x.clone(), rest[:4], rest[4], None if rest[4] is None else rest[4].clone()))
    orig_add(self, dout, x, dw, *rest)
F_.DeferredWgrads.add = add
orig_flush = F_.DeferredWgrads.flush
def flush(self):
    torch.cuda.synchronize()
    for i, (d, dc, x, xc, shape, sc, scc) in enumerate(snaps):
        dd = float((d.float() - dc.float()).abs().max()); dx = float((x.float() - xc.float()).abs().max())
        ds = 0.0 if sc is None else float((sc - scc).abs().max())
        if dd or dx or ds:
            print("item", i, shape, "dout changed", dd, "x changed", dx, "scale changed", ds, "dout ptr", d.data_ptr(), "x ptr", x.data_ptr())
    print("checked", len(snaps))
    orig_flush(self)
F_.DeferredWgrads.flush = flush
F_.DEFER_WGRAD = 1
opt.zero_grad()
_, _, z_i, z_j = model(x_i.cuda(), x_j.cuda())
ntxent_loss(z_i, z_j, GRAFP_CFG).backward()
torch.cuda.synchronize()
