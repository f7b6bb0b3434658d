import sys, os, torch
sys.path.insert(0, os.getcwd())
from bench import CFG, synth_clips
from neuralsampleid_amd import functional as F_, ops, fingerprint
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
ops.set_gemm_precision("bf16"); F_.set_activation_dtype("bf16")
torch.manual_seed(42)
model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=3, size="t")).cuda().eval()
x,_ = synth_clips(2048, 1, "cuda")
out = torch.empty((2048, CFG["d"]), device="cuda")
for _ in range(2): fingerprint.extract_fingerprints(model, x, 2048, out)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fingerprint.extract_fingerprints(model, x, 2048, out)
    torch.cuda.synchronize()
evs=[e for e in prof.events() if e.name in ("aten::copy_","aten::clone","aten::contiguous","aten::to","aten::_to_copy","aten::zeros","aten::fill_","aten::empty_like")]
from collections import Counter
c=Counter()
for e in evs:
    st=[s for s in (e.stack or []) if "neuralsampleid_amd" in s or "bench" in s]
    c[(e.name, tuple(st[:3]))]+=1
for k,v in c.most_common(30): print(v, k)
