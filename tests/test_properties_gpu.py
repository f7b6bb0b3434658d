"""GPU: size-independent properties at the sizes BASELINE.json names, where no reference output exists to compare with (the oracle takes
minutes per micro-batch there): config 5's 2 048-clip extraction micro-batch and config 3's global batch of 2 048 pairs.

  * kNN (torch_edge.py:70-103, 270-284) at 2 048 clips, every stage shape: self first; the returned neighbours in non-decreasing fp64
    distance; the search is per clip — a permutation of the clips permutes the result, bit for bit.
  * max-relative aggregation (torch_vertex.py:21-32) at 2 048 clips: even channels are the input, odd channels are >= 0 when the node is
    in its own list, the arg-max bytes name a neighbour that attains the maximum, and the clips do not see each other.
  * extraction (generate.py:42-46): eval-mode embeddings are per clip — 2 048 clips in one micro-batch (gemm256, knn2_pair, the
    register-tile FFN at every width) against the same clips in four micro-batches of 512 (other launch thresholds) and in another order
    (equal up to the head's atomics noise); unit norm.
  * NT-Xent (ntxent.py:5-30) at B = 2 048: swapping the views swaps the gradients and keeps the loss; a permutation of the pairs permutes
    the gradients; the loss lies in [0, log(2B - 1) + 2 / tau] and equals an fp64 evaluation of the closed form (a 4 096 x 4 096
    matrix is cheap on the CPU: the one reference-arithmetic check that does exist at this size)."""
import math

import pytest
import torch

from synth import GRAFP_CFG, synth_randn

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture()
def bf16_mode():
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops
    ops.set_gemm_precision("bf16")
    F_.set_activation_dtype("bf16")
    yield ops
    ops.set_gemm_precision("fp32")           # the suite's default mode
    F_.set_activation_dtype("fp32")


@pytest.mark.parametrize("N,C", [(256, 64), (128, 128), (64, 256), (32, 512)])
def test_knn_at_the_extraction_micro_batch(bf16_mode, N, C):
    ops = bf16_mode
    B, k = 2048, 3
    y = synth_randn(f"propknn{N}", B * N, C).to(BF).to(DEV)
    idx = ops.knn_graph(y, B, N, C, k, 1, None)
    assert idx.shape == (B, N, k) and int(idx.min()) >= 0 and int(idx.max()) < N
    assert bool((idx[..., 0] == torch.arange(N, device=DEV)).all())                       # self first
    yn = y.double().reshape(B, N, C)
    yn = yn / yn.norm(dim=2, keepdim=True).clamp_min(1e-12)
    nb = torch.gather(yn.unsqueeze(1).expand(B, N, N, C), 2, idx.long().unsqueeze(-1).expand(B, N, k, C))
    d = ((nb - yn.unsqueeze(2)) ** 2).sum(-1)                                             # (B, N, k) fp64 distances of the returned ids
    assert bool((d[..., 1:] - d[..., :-1] >= -2e-6).all())                                # non-decreasing up to the split product's error
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(N)).to(DEV)
    idx_p = ops.knn_graph(y.reshape(B, N, C)[perm].reshape(B * N, C).contiguous(), B, N, C, k, 1, None)
    assert torch.equal(idx_p, idx[perm])                                                  # per clip, deterministic


@pytest.mark.parametrize("N,C,k", [(256, 64, 3), (64, 256, 18)])
def test_max_relative_aggregation_at_the_extraction_micro_batch(bf16_mode, N, C, k):
    ops = bf16_mode
    B = 2048
    y = synth_randn(f"propmr{N}", B * N, C).to(BF).to(DEV)
    g = torch.Generator().manual_seed(k)
    idx = torch.randint(0, N, (B, N, k), generator=g).to(torch.int32)
    idx[..., 0] = torch.arange(N)                                                          # the node itself is in its list (as kNN returns)
    idx = idx.to(DEV)
    u, am = ops.mr_aggregate_fwd(y, idx, B, N, C)
    u3 = u.reshape(B * N, C, 2)
    assert torch.equal(u3[..., 0], y)                                                      # even channels: the input
    assert float(u3[..., 1].float().min()) >= 0.0                                          # max over a set that holds d = 0
    tgt = torch.gather(idx.long(), 2, am.reshape(B, N, C).long())                          # the neighbour the arg-max byte names
    rows = (torch.arange(B, device=DEV).view(B, 1, 1) * N + tgt).reshape(B * N, C)
    picked = torch.gather(y.float(), 0, rows) - y.float()
    assert torch.equal(picked.to(BF), u3[..., 1])                                          # ... attains the stored maximum
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    u_p, am_p = ops.mr_aggregate_fwd(y.reshape(B, N, C)[perm].reshape(B * N, C).contiguous(), idx[perm].contiguous(), B, N, C)
    assert torch.equal(u_p.reshape(B, N, 2 * C), u.reshape(B, N, 2 * C)[perm]) and torch.equal(am_p.reshape(B, N, C), am.reshape(B, N, C)[perm])


def test_extraction_is_per_clip_at_the_timed_micro_batch(bf16_mode):
    from neuralsampleid_amd import fingerprint
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    ops = bf16_mode
    torch.manual_seed(42)
    model = SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=3, size="t")).to(DEV).eval()
    clips = synth_randn("propclips", 2048, 64 * 128).reshape(2048, 64, 128).to(DEV)
    ops.launch_counters(reset=True)
    z_all = fingerprint.extract_fingerprints(model, clips, 2048)
    cnt = ops.launch_counters()
    assert cnt["gemm256"] > 0 and cnt["knn2_raw"] > 0 and cnt["ffn_fused"] == 10 and cnt["mrconv_fused"] == 10, cnt       # (C = 512: unfused)
    z_parts = fingerprint.extract_fingerprints(model, clips, 512)
    torch.cuda.synchronize()
    assert float((z_all.norm(dim=1) - 1).abs().max()) < 1e-5                               # F.normalize
    cos = torch.nn.functional.cosine_similarity(z_all, z_parts, dim=1)
    assert float(cos.min()) > 0.9995, float(cos.min())                                     # other tile shapes / launch thresholds, same clips
    perm = torch.randperm(2048, generator=torch.Generator().manual_seed(3)).to(DEV)
    z_perm = fingerprint.extract_fingerprints(model, clips[perm].contiguous(), 2048)
    # no clip sees another; not bit for bit: the projector head's split-K partial sums meet in fp32 atomics (tools/forward_soak.py:
    # run-to-run deviation 7e-8 .. 9e-8 on unit-norm embeddings)
    assert float((z_perm - z_all[perm]).abs().max()) < 1e-6, float((z_perm - z_all[perm]).abs().max())


def test_ntxent_at_the_global_batch():
    from neuralsampleid_amd import ops
    B, d, tau = 2048, 128, 0.05
    z_i = torch.nn.functional.normalize(synth_randn("propzi", B, d), dim=1).to(DEV)
    z_j = torch.nn.functional.normalize(z_i.cpu() + 0.3 * synth_randn("propzj", B, d), dim=1).to(DEV)
    loss, dzi, dzj = ops.ntxent_fwd_bwd(z_i, z_j, tau)
    loss_s, dzi_s, dzj_s = ops.ntxent_fwd_bwd(z_j, z_i, tau)                               # views swapped
    assert abs(float(loss) - float(loss_s)) < 1e-6 * max(1.0, abs(float(loss)))
    assert float((dzi - dzj_s).abs().max()) < 1e-6 and float((dzj - dzi_s).abs().max()) < 1e-6
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9)).to(DEV)
    loss_p, dzi_p, dzj_p = ops.ntxent_fwd_bwd(z_i[perm].contiguous(), z_j[perm].contiguous(), tau)
    assert abs(float(loss) - float(loss_p)) < 1e-6 * max(1.0, abs(float(loss)))
    assert float((dzi_p - dzi[perm]).abs().max()) < 1e-6 and float((dzj_p - dzj[perm]).abs().max()) < 1e-6
    assert 0.0 <= float(loss) <= math.log(2 * B - 1) + 2.0 / tau
    # the closed form on the CPU in fp64 (a 4 096 x 4 096 matrix: cheap) — the one reference-arithmetic check that does exist at this size
    z = torch.stack((z_i.cpu().double(), z_j.cpu().double()), 1).reshape(2 * B, d)
    a = (z @ z.t()) / tau
    a.fill_diagonal_(float("-inf"))
    ref = (torch.logsumexp(a, 1) - a[torch.arange(2 * B), torch.arange(2 * B) ^ 1]).mean()
    assert abs(float(loss) - float(ref)) < 2e-5
