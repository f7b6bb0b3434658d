"""Block-level forward/backward of the Grapher/FFN stack, composed from the HIP ops (ops.py).

Layout: node-major rows (B*N, C).  A conv+BN(+act) layer keeps only its RAW conv output `r`; the BatchNorm
affine (scale, shift) rides along as a `BNAffine` and is applied by whichever kernel consumes `r` next (the next
GEMM's operand load, the kNN/aggregation kernels, or `bn_apply` where the residual stream is materialised).

Every `*_forward(x, P, S, ...)` takes the parameters as a dict `P` of tensors named after the reference's
state_dict suffixes, and stashes what backward needs in the plain dict `S` (None: inference, nothing kept).
Every `*_backward(dout, P, S, G)` accumulates parameter gradients INTO the tensors of `G` (same keys as P) and
returns the input gradient.  The autograd wrappers below and the fused training step share these functions.

Reference sites: Grapher.forward encoder/gcn_lib/torch_vertex.py:183-195, DyGraphConv2d :126-139, MRConv2d :19-34,
FFN.forward encoder/graph_encoder.py:82-89, Downsample :48-50, GraphEncoder.forward :190-214,
SimCLR.forward simclr/simclr.py:31-47, GPUPeakExtractorv2.forward peak_extractor.py:45-70.
"""
from typing import Dict, List, Optional

import torch

from . import ops
from .ops import ACT_ELU, ACT_LEAKY, ACT_NONE, ACT_RELU, BNAffine

Tensor = torch.Tensor


class KnnTape:
    """Debug/test hook: record the neighbour indices of every kNN call, optionally replacing them (teacher forcing).
    kNN is discontinuous, so end-to-end parity is stated as: indices equal outside near-ties, outputs equal when the
    indices are forced (tests/test_e2e_gpu.py)."""

    def __init__(self, replay: Optional[List[Tensor]] = None, cyclic: bool = False, patch=None):
        """cyclic: the replay list describes ONE step and is re-used by every following step (warm-up steps and the
        capture of a hipGraph: the captured kernels then read the replayed index tensors on every replay).
        patch: per call (flat row numbers, ids for those rows) — the search's own result is kept except on those rows (a fixture
        that stores the reference's ids only where the distance margin is too small to call: tests/b256_common.py); the
        graphs actually used are collected in .patched"""
        self.replay = list(replay) if replay is not None else None
        self.recorded: List[Tensor] = []
        self.pos = 0
        self.cyclic = cyclic
        self.patch = list(patch) if patch is not None else None
        self.patched: List[Tensor] = []


TAPE: Optional[KnnTape] = None

# A conv bias that feeds a training-mode BatchNorm directly has an analytically ZERO gradient (BN subtracts the batch
# mean, so sum_m dL/dr[m, c] == 0).  The reference's autograd still evaluates that sum and gets +-1e-7 roundoff, which
# Adam then normalises into a +-lr random walk of a parameter with no effect on the output.  By default the column sum
# is skipped and the gradient left at exactly 0; setting functional.EXACT_BIAS_GRAD = True evaluates it like the reference does.
EXACT_BIAS_GRAD = False

# When every parameter of a block already owns a gradient buffer and this flag is set (optim.FusedClipAdam sets it),
# backward accumulates straight into p.grad (views of the flat gradient buffer) instead of returning fresh tensors
# for autograd to add: saves one zero-fill and one add per parameter per view.
DIRECT_GRADS = False

# Storage type of activation tensors (features, raw conv outputs and their gradients) from the pipeline entry onwards
# (peak-extractor output / layout conversion). torch.float32 (default, strict parity) or torch.bfloat16 (BASELINE
# config 2: bf16 storage, fp32 accumulate — halves the HBM traffic of a step). Parameters, statistics, indices and the
# small head tensors stay fp32 either way. Set through set_activation_dtype().
ACT_DTYPE = torch.float32


def set_activation_dtype(dtype) -> None:
    global ACT_DTYPE
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(dtype, dtype)
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("activation storage is torch.float32 or torch.bfloat16")
    ACT_DTYPE = dtype


class ViewOrder:
    """Two-stream execution of the two views (simclr.SimCLR(overlap_views=True)): the views are independent until
    NT-Xent, so view j runs on a side stream and its kernels fill the launch gaps / low-occupancy tails of view i's.
    The only cross-view state in forward is the BatchNorm running statistics, which the reference updates view i
    first, then view j (simclr.py:36,42). While `mode` is "a" / "b" the finalize kernels leave the running statistics
    alone and every layer's (running tensors, batch mean, unbiased variance) is noted per view; after both forwards
    SimCLR applies all updates in that order with ops.bn_running_update — the same float expressions in the same order,
    and no cross-stream edge in the captured graph (per-layer events between the branches cost 0.2 ms per step).
    (In backward every shared accumulation is atomic.)"""

    def __init__(self):
        self.mode, self.pending = None, {"a": [], "b": []}


VIEW_ORDER = ViewOrder()
SIDE_STREAMS = []      # side streams that carry a view's forward AND backward kernels (registered by SimCLR)


def join_side_streams() -> None:
    """Make the current stream wait for everything enqueued on the side streams. With direct gradient accumulation
    there are no AccumulateGrad nodes, so autograd does not join a side stream at the end of backward: whoever consumes
    the gradients next (optimiser step, gradient all-reduce) must."""
    if SIDE_STREAMS:
        cur = torch.cuda.current_stream()
        for st in SIDE_STREAMS:
            if st != cur:
                cur.wait_stream(st)


class BwdLink:
    """Hand-over between two consecutive blocks of one view: the producer block materialised its output with
    bn_apply(r, aff, act) (+ residual); the consumer block's LAST backward GEMM writes the gradient of that output, i.e. the
    dL/dy of the producer's BatchNorm — so it also emits that BatchNorm-backward's column sums (ops.linear_bwd_data(bn=...))
    and the producer's backward skips its reduce launch."""
    __slots__ = ("bn", "partial")

    def __init__(self, bn):
        self.bn, self.partial = bn, None


class BwdChain:
    """links of the blocks of one encoder forward, in forward order (set by GraphEncoder.forward_rows while it runs)"""

    def __init__(self):
        self.last = None

    def produce(self, S, r, aff, act):
        if S is not None:
            S["link_out"] = self.last = BwdLink((r, aff, act))

    def consume(self, S):
        if S is not None:
            S["link_in"] = self.last


CHAIN = None          # BwdChain of the encoder forward in progress (None: blocks run stand-alone, nothing is linked)


def _link_in(S):
    link = S.get("link_in")
    return link.bn if link is not None else None


def _link_partial(S):
    link = S.get("link_out")
    if link is None:
        return None
    part, link.partial = link.partial, None
    return part


def _store_partial(S, part):
    link = S.get("link_in")
    if link is not None:
        link.partial = part


# Called at the end of every block backward with the parameter tensors whose gradient contribution has just been
# enqueued (parallel.GradReducer uses it to overlap the bucketed all-reduce with the rest of backward).
GRAD_READY_HOOK = None


# ------------------------------------------------------------------------------------------------ deferred weight gradients
# Nothing waits for a weight gradient until the optimiser (or the gradient all-reduce), yet launched in stream order it sits on its
# view's dependent chain: the next backward-data GEMM starts behind it. With DEFER_WGRAD the conv layers of the Grapher / FFN blocks, the
# stem, the Downsample layers and the head only RECORD their weight-gradient problem (dr, the saved GEMM input and its pending affine stay
# alive); when autograd has run the whole backward pass (engine callback) the two view streams are joined ONCE and the recorded
# problems are issued together (ops.linear_bwd_weight_batch: both views of a layer as one problem over 2M rows, the layers grouped into
# one launch per tile class), on two lanes.
# References: the backward of every conv at encoder/gcn_lib/torch_vertex.py:152-162, encoder/graph_encoder.py:44,74-77,
# encoder/gcn_lib/torch_nn.py:56, simclr/simclr.py:25-28; train.py:70-75.
DEFER_WGRAD = 1      # one-box A/B of the whole step (round 6, x2): 7.73 in-chain -> 8.68 deferred as per-layer launches -> 7.36 ms grouped
DEFER_TWO_LANES = 1  # the short launches of the phase on the second view stream beside the two big ones: 7.40 -> 7.29 ms (x3); moving
#                      half of the wide problems over as well: equal (7.284 / 7.275)
DEFER_CHUNKS = 1     # pieces of the deferred phase when a gradient-ready hook is installed (parallel.GradReducer.install sets 3 under data parallelism)
# (Starting the phase EARLY on a third stream beside the rest of backward was built and measured slower at every setting: +0.15 ... +0.6 ms;
#  the code is tools/variants/deferred_fork_functional.py, the numbers docs/experiments.md.)


class DeferredWgrads:
    def __init__(self):
        self.items, self.hooks, self.calls, self.armed = [], [], [], False
        self.verify = None      # tests: a list that receives (dw, the same sum through the per-layer launches) for every layer of a flush

    def reset(self):
        """drop whatever an interrupted backward pass left behind (optim.FusedClipAdam.zero_grad calls it at the start of every step)"""
        self.items, self.hooks, self.calls, self.armed = [], [], [], False

    def _arm(self):
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add(self, dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in, extra=None):
        self.items.append((dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in, extra))
        self._arm()

    def add_call(self, fn, tensors):
        """other work that only produces parameter gradients (the peak extractor's backward, bias column sums): runs on the phase's second lane"""
        self.calls.append((fn, tensors))
        self._arm()

    def note_hook(self, params, n_before=0):
        """the block whose backward has just recorded the items [n_before:] reports its parameters when all of those have been issued"""
        self.hooks.append((params, {it[2].data_ptr() for it in self.items[n_before:]}))

    def _run_calls(self):
        calls, self.calls = self.calls, []
        for fn, tensors in calls:
            for t in tensors:
                t.record_stream(torch.cuda.current_stream())
            fn()

    def _issue(self, items):
        if not items:
            return
        for it in items:                   # the launch reads tensors that were produced on the view streams
            it[0].record_stream(torch.cuda.current_stream())
            it[1].record_stream(torch.cuda.current_stream())
        ops.linear_bwd_weight_batch(items)
        if self.verify is not None:     # tests: the same problems through the per-layer launches, from the tensors as they are NOW
            tmp = {}
            for it in items:
                t = tmp.setdefault(it[2].data_ptr(), (it[2], torch.zeros_like(it[2])))[1]
                ops.wgrad_item(it[:2] + (t,) + it[3:])
            self.verify.extend((dw, t) for dw, t in tmp.values())

    def _issue_two_lanes(self, items):
        """The wide layers' problems (the 8-wave 128x128 classes: ~90 % of the flops, two launches, HBM-bound) on the current stream and
        everything else (five short launches that fill the chip badly: the C = 64 / 128 layers, Downsample, stem, the fp32 head — 0.35 ms
        of the 1.37 ms phase in an eager trace — and the recorded calls) on the idle view stream BESIDE them: no dependency chain on
        either side, so unlike a launch beside backward the co-resident kernels only fill each other's gaps. One fork edge, one join edge."""
        side = SIDE_STREAMS[0] if SIDE_STREAMS else None
        heavy = [it for it in items if it[0].dtype == torch.bfloat16 and it[10] is None and it[3] % 128 == 0 and it[4] % 128 == 0 and
                 it[5] % 128 == 0]
        if not DEFER_TWO_LANES or side is None or side == torch.cuda.current_stream() or not heavy or \
                (len(heavy) == len(items) and not self.calls):
            self._issue(items)
            self._run_calls()
            return
        light = [it for it in items if not any(it is h for h in heavy)]
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._issue(light)
            self._run_calls()
        self._issue(heavy)
        main.wait_stream(side)

    def flush(self):
        """runs on the thread that called backward(), once every backward node has been enqueued"""
        self.armed = False
        items, hooks, self.items, self.hooks = self.items, self.hooks, [], []
        if not items and not hooks and not self.calls:
            return
        join_side_streams()
        chunks = max(1, int(DEFER_CHUNKS)) if GRAD_READY_HOOK is not None else 1
        if chunks == 1:
            self._issue_two_lanes(items)
            if GRAD_READY_HOOK is not None:
                for params, _ in hooks:
                    GRAD_READY_HOOK(params)
            return
        # Data parallelism: the phase goes out in `chunks` pieces of about equal gradient size, in backward order (late layers first:
        # the order the reducer cuts its buckets in), and every block reports its parameters as soon as its piece has been issued -- the
        # bucketed all-reduce of piece c runs on the communicator's stream beside the launches of piece c + 1.
        order, size = [], {}
        for it in items:
            ptr = it[2].data_ptr()
            if ptr not in size:
                order.append(ptr)
                size[ptr] = it[2].numel()
        total, acc, piece_of = float(sum(size.values())), 0.0, {}
        for ptr in order:
            piece_of[ptr] = min(chunks - 1, int(acc * chunks / total))
            acc += size[ptr]
        fired = [False] * len(hooks)
        for c in range(chunks):
            if c == 0:
                self._run_calls()          # (before any hook fires: a hook without recorded problems reports at the first piece)
            self._issue([it for it in items if piece_of[it[2].data_ptr()] == c])
            for i, (params, ptrs) in enumerate(hooks):
                if not fired[i] and all(piece_of.get(p_, 0) <= c for p_ in ptrs):
                    fired[i] = True
                    GRAD_READY_HOOK(params)


DEFERRED = DeferredWgrads()


def _wgrad(dout, x, dw, M, Nout, K, groups=1, in_scale=None, in_shift=None, act_in=ACT_NONE):
    """dw += dout^T f(x): now, or recorded for the deferred phase (only inside a backward pass that accumulates straight into p.grad:
    an autograd-returned gradient tensor must be complete when its node returns)"""
    if _deferring() and (DEFER_HEAD or dout.dtype == torch.bfloat16):
        DEFERRED.add(dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in)
    else:
        ops.linear_bwd_weight(dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in)


_IN_DIRECT_BACKWARD = False     # set by _BlockFn.backward while a block accumulates into p.grad


DEFER_DS = 1        # the Downsample layers' packed weight gradients join the deferred phase
DEFER_HEAD = 1      # so do the fp32 problems of the projection / projector head (256 rows each: tile-count bound launches at the turnaround)


def _deferring() -> bool:
    """bf16 storage only (the projector head's fp32 tensors included): the strict-fp32 path keeps its in-chain launches"""
    return bool(DEFER_WGRAD and DIRECT_GRADS and _IN_DIRECT_BACKWARD and ACT_DTYPE == torch.bfloat16)


def _bias_grad(dout, g):
    """g += column sums of dout (a Linear's bias gradient): with the deferred phase a recorded call on its second lane (three 5 us
    launches per view at the turnaround between forward and backward, where nothing else runs)"""
    if _deferring() and DEFER_HEAD:
        DEFERRED.add_call(lambda: ops.colsum_acc(dout, g), (dout,))
    else:
        ops.colsum_acc(dout, g)


def _bias_grad_before_bn(dr, g):
    if EXACT_BIAS_GRAD:
        ops.colsum_acc(dr, g)


def _bn(P, S, pre):
    """the BatchNorm layer's tensors, and whether the block runs without a backward (S is None): only then may an eval-mode
    BatchNorm be folded into its conv"""
    return P[pre + "weight"], P[pre + "bias"], P[pre + "running_mean"], P[pre + "running_var"], \
        P.get(pre + "num_batches_tracked"), S is None


# Eval mode with bf16 storage (fingerprint extraction): every BatchNorm is folded into the conv in front of it
# (ops.folded_conv_bn), so a conv+BN layer is ONE GEMM that writes normalised values, the residual shortcut rides in that GEMM's
# epilogue (nsid_linear_fwd_res) and the bn_apply passes disappear. The strict-fp32 path keeps conv and BatchNorm apart,
# as the parity tests state them.
FOLD_EVAL_BN = True


def fold_eval(training: bool, S) -> bool:
    """S: the block's saved-tensor dict (None = no gradient is needed). Eval mode WITH gradients keeps the unfolded path."""
    return (not training) and S is None and FOLD_EVAL_BN and ACT_DTYPE == torch.bfloat16


def conv_bn(x: Tensor, M: int, K: int, Nout: int, w: Tensor, bias: Optional[Tensor], bn, training: bool,
            groups: int = 1, in_aff: Optional[BNAffine] = None, act_in: int = ACT_NONE, residual: Optional[Tensor] = None,
            folded_act: int = ACT_NONE):
    """r = f(x) W^T (+b) on MFMA, plus the affine of the BatchNorm that follows (batch stats when training).
    residual: only when the BatchNorm is folded (eval mode, no backward) — the result is then BN(r) + residual, with the
    identity affine.
    folded_act: the activation that FOLLOWS this layer's BatchNorm (BasicConv's / FFN's ReLU). Used only on the folded path, where
    the GEMM writes act(BN(conv)) itself (ReLU commutes with the bf16 rounding of the store: same stored values as a ReLU on
    load) and returns None as the affine: the consumer then loads plain values (in_aff = None, act_in = ACT_NONE)."""
    gamma, beta, rm, rv, nbt, nograd = bn
    if nograd and fold_eval(training, None):
        wf, bf = ops.folded_conv_bn(ops.w2d(w), bias, gamma, beta, rm, rv)
        sc, sh = (in_aff.scale, in_aff.shift) if in_aff else (None, None)
        if in_aff is not None and in_aff.identity and act_in == ACT_RELU and M % 256 == 0 and Nout % 128 == 0 and \
                K % 64 == 0 and wf.numel() % 8 == 0:
            sc = sh = None       # the producer's BatchNorm is folded too: ReLU alone, on the packed bf16 values
        if in_aff is None:
            act_in = ACT_NONE
        r, _ = ops.linear_fwd(x, wf, bf, M, Nout, K, groups, sc, sh, act_in, folded_act, addend=residual)
        return r, (None if folded_act != ACT_NONE else ops.identity_affine(groups * Nout, x.device))
    assert residual is None
    r, stat = ops.linear_fwd(x, ops.w2d(w), bias, M, Nout, K, groups,
                             in_aff.scale if in_aff else None, in_aff.shift if in_aff else None, act_in,
                             ACT_NONE, want_stat=training)
    return r, bn_affine_from(stat, M, gamma, beta, rm, rv, nbt, training)


def bn_affine_from(stat, M, gamma, beta, rm, rv, nbt, training: bool) -> BNAffine:
    """the affine of the BatchNorm that follows a conv: batch statistics (from the GEMM epilogue's partial sums) when
    training, running statistics otherwise"""
    if training:
        vo = VIEW_ORDER
        if vo.mode is not None and rm is not None:
            aff, uvar = ops.bn_finalize_deferred(stat, M, gamma, beta)
            vo.pending[vo.mode].append((rm, rv, nbt, aff.mean, uvar))
            return aff
        return ops.bn_finalize(stat, M, gamma, beta, rm, rv, nbt)
    return ops.bn_eval_affine(gamma, beta, rm, rv)


# ------------------------------------------------------------------------------------------------ stem
def stem_forward(nodes: Tensor, P: Dict[str, Tensor], S: Optional[dict], training: bool) -> Tensor:
    M, K = nodes.shape
    C = P["0.weight"].shape[0]
    r, aff = conv_bn(nodes, M, K, C, P["0.weight"], None, _bn(P, S, "1."), training)
    x0 = ops.bn_apply(r, aff, ACT_LEAKY)
    if S is not None:
        S.update(nodes=nodes, r=r, aff=aff)
        if CHAIN is not None:
            CHAIN.produce(S, r, aff, ACT_LEAKY)
    return x0


def stem_backward(dx0: Tensor, P, S, G, need_input_grad: bool = True) -> Optional[Tensor]:
    nodes, r, aff = S["nodes"], S["r"], S["aff"]
    M, K = nodes.shape
    C = r.shape[1]
    dr = ops.bn_backward(dx0, r, aff, ACT_LEAKY, G["1.weight"], G["1.bias"], partial=_link_partial(S))
    _wgrad(dr, nodes, ops.w2d(G["0.weight"]), M, C, K)
    return ops.linear_bwd_data(dr, ops.w2d(P["0.weight"]), M, C, K) if need_input_grad else None


# ------------------------------------------------------------------------------------------------ Grapher
def grapher_forward(x0: Tensor, P, S: Optional[dict], B: int, N: int, k: int, dilation: int, training: bool) -> Tensor:
    M, C = x0.shape
    r1, a1 = conv_bn(x0, M, C, C, P["fc1.0.weight"], P["fc1.0.bias"], _bn(P, S, "fc1.1."), training)
    idx = ops.knn_graph(r1, B, N, C, k, dilation, a1)
    if TAPE is not None:
        TAPE.recorded.append(idx)
        if TAPE.replay is not None:
            idx = TAPE.replay[TAPE.pos].to(device=idx.device, dtype=torch.int32).contiguous()
            TAPE.pos = (TAPE.pos + 1) % len(TAPE.replay) if TAPE.cyclic else TAPE.pos + 1
        elif TAPE.patch is not None:
            rows, ids = TAPE.patch[TAPE.pos]
            TAPE.pos += 1
            idx = idx.clone()
            if len(rows):
                idx.view(-1, idx.shape[-1])[torch.as_tensor(rows, device=idx.device, dtype=torch.long)] = \
                    torch.as_tensor(ids, device=idx.device).to(idx.dtype)
            TAPE.patched.append(idx)
    pre = "graph_conv.gconv.nn."
    r2 = None
    if fold_eval(training, S) and a1 is not None and a1.identity:
        # forward-only, BatchNorms folded: aggregation + grouped conv in one launch per clip, the interleaved tensor u never formed
        gg, gb, grm, grv, _, _ = _bn(P, S, pre + "1.")
        wgf, bgf = ops.folded_conv_bn(ops.w2d(P[pre + "0.weight"]), P[pre + "0.bias"], gg, gb, grm, grv)
        r2, a2 = ops.mrconv_fused_fwd(r1, idx, B, N, C, wgf, bgf), None
    if r2 is None:
        u, amax = ops.mr_aggregate_fwd(r1, idx, B, N, C, a1, want_argmax=S is not None)
        r2, a2 = conv_bn(u, M, C // 2, C // 2, P[pre + "0.weight"], P[pre + "0.bias"], _bn(P, S, pre + "1."), training,
                         groups=4, folded_act=ACT_RELU)
    if fold_eval(training, S):   # conv + BatchNorm + shortcut in one launch; r2 already is relu(BN(conv)) (a2 is None)
        return conv_bn(r2, M, 2 * C, C, P["fc2.0.weight"], P["fc2.0.bias"], _bn(P, S, "fc2.1."), training,
                       in_aff=a2, act_in=ACT_RELU, residual=x0)[0]
    r3, a3 = conv_bn(r2, M, 2 * C, C, P["fc2.0.weight"], P["fc2.0.bias"], _bn(P, S, "fc2.1."), training,
                     in_aff=a2, act_in=ACT_RELU)
    x1 = ops.bn_apply(r3, a3, ACT_NONE, residual=x0)
    if S is not None:
        S.update(x0=x0, r1=r1, a1=a1, idx=idx, amax=amax, u=u, r2=r2, a2=a2, r3=r3, a3=a3, B=B, N=N)
        if CHAIN is not None:
            CHAIN.consume(S)                       # x0 was materialised by the previous block's BatchNorm
            CHAIN.produce(S, r3, a3, ACT_NONE)
    return x1


def grapher_backward(dx1: Tensor, P, S, G) -> Tensor:
    x0, r1, a1, idx, amax, u, r2, a2, r3, a3 = (S[k_] for k_ in ("x0", "r1", "a1", "idx", "amax", "u", "r2", "a2",
                                                                  "r3", "a3"))
    B, N = S["B"], S["N"]
    M, C = x0.shape
    pre = "graph_conv.gconv.nn."
    # fc2 (+BN), input = relu(BN(r2))
    # dv is dL/d(relu(BN(r2))): the GEMM that writes it also emits that BatchNorm's backward column sums; where the weight-stationary
    # form serves the shape (ops.WS_BWD_SHAPES) it evaluates fc2's own BatchNorm backward on its operand load too (no apply pass)
    dr3, dv, part2 = ops.bn_backward_linear_bwd_data(dx1, r3, a3, ACT_NONE, G["fc2.1.weight"], G["fc2.1.bias"], _link_partial(S),
                                                     ops.w2d(P["fc2.0.weight"]), M, C, 2 * C, 1, bn=(r2, a2, ACT_RELU),
                                                     site=ops.SITE_FC2, inplace=False)
    _bias_grad_before_bn(dr3, G["fc2.0.bias"])
    _wgrad(dr3, r2, ops.w2d(G["fc2.0.weight"]), M, C, 2 * C, 1, a2.scale, a2.shift, ACT_RELU)
    # grouped conv (+BN+ReLU), input = u
    # (the BatchNorm-backward apply is evaluated on the backward-data GEMM's operand load where the shape allows: ops.py)
    dr2, du, _ = ops.bn_backward_linear_bwd_data(dv, r2, a2, ACT_RELU, G[pre + "1.weight"], G[pre + "1.bias"], part2,
                                                 ops.w2d(P[pre + "0.weight"]), M, C // 2, C // 2, 4, site=ops.SITE_GCONV)
    _bias_grad_before_bn(dr2, G[pre + "0.bias"])
    _wgrad(dr2, u, ops.w2d(G[pre + "0.weight"]), M, C // 2, C // 2, 4)
    # max-relative aggregation: route to arg-max neighbour and centre; kNN itself carries no gradient
    # (bf16 storage: the launch also emits the backward column sums of fc1's BatchNorm, one row per clip — no reduce launch)
    dy, part1 = ops.mr_aggregate_bwd(du, idx, amax, B, N, C, bn=(r1, a1, ACT_NONE))
    # fc1 (+BN), input = x0
    dr1, dx0, part = ops.bn_backward_linear_bwd_data(dy, r1, a1, ACT_NONE, G["fc1.1.weight"], G["fc1.1.bias"], part1,
                                                     ops.w2d(P["fc1.0.weight"]), M, C, C, 1, addend=dx1, bn=_link_in(S) or False,
                                                     site=ops.SITE_FC1)
    _bias_grad_before_bn(dr1, G["fc1.0.bias"])
    _wgrad(dr1, x0, ops.w2d(G["fc1.0.weight"]), M, C, C)
    _store_partial(S, part)
    return dx0


# ------------------------------------------------------------------------------------------------ FFN
def ffn_forward(x1: Tensor, P, S: Optional[dict], training: bool) -> Tensor:
    M, C = x1.shape
    H = P["fc1.0.weight"].shape[0]
    if fold_eval(training, S) and C in (64, 128, 256) and H == 4 * C and M % (256 if C == 256 else 128) == 0:
        # forward-only, both BatchNorms folded into their convs: one launch, the hidden tensor never leaves the CU (csrc/ffn_fused.hip)
        g1, be1, rm1, rv1, _, _ = _bn(P, S, "fc1.1.")
        g2, be2, rm2, rv2, _, _ = _bn(P, S, "fc2.1.")
        w1f, b1f = ops.folded_conv_bn(ops.w2d(P["fc1.0.weight"]), None, g1, be1, rm1, rv1)
        w2f, b2f = ops.folded_conv_bn(ops.w2d(P["fc2.0.weight"]), None, g2, be2, rm2, rv2)
        out = ops.ffn_fused_fwd(x1, w1f, b1f, w2f, b2f, M, C, H)
        if out is not None:
            return out
    r4, a4 = conv_bn(x1, M, C, H, P["fc1.0.weight"], None, _bn(P, S, "fc1.1."), training, folded_act=ACT_RELU)
    if fold_eval(training, S):     # r4 already is relu(BN(conv)) (a4 is None)
        return conv_bn(r4, M, H, C, P["fc2.0.weight"], None, _bn(P, S, "fc2.1."), training, in_aff=a4, act_in=ACT_RELU,
                       residual=x1)[0]
    r5, a5 = conv_bn(r4, M, H, C, P["fc2.0.weight"], None, _bn(P, S, "fc2.1."), training, in_aff=a4, act_in=ACT_RELU)
    x2 = ops.bn_apply(r5, a5, ACT_NONE, residual=x1)
    if S is not None:
        S.update(x1=x1, r4=r4, a4=a4, r5=r5, a5=a5)
        if CHAIN is not None:
            CHAIN.consume(S)
            CHAIN.produce(S, r5, a5, ACT_NONE)
    return x2


def ffn_backward(dx2: Tensor, P, S, G) -> Tensor:
    x1, r4, a4, r5, a5 = S["x1"], S["r4"], S["a4"], S["r5"], S["a5"]
    M, C = x1.shape
    H = r4.shape[1]
    dr5, dh, part4 = ops.bn_backward_linear_bwd_data(dx2, r5, a5, ACT_NONE, G["fc2.1.weight"], G["fc2.1.bias"], _link_partial(S),
                                                     ops.w2d(P["fc2.0.weight"]), M, C, H, 1, bn=(r4, a4, ACT_RELU), site=ops.SITE_FFN2,
                                                     inplace=False)
    _wgrad(dr5, r4, ops.w2d(G["fc2.0.weight"]), M, C, H, 1, a4.scale, a4.shift, ACT_RELU)
    dr4, dx1, part = ops.bn_backward_linear_bwd_data(dh, r4, a4, ACT_RELU, G["fc1.1.weight"], G["fc1.1.bias"], part4,
                                                     ops.w2d(P["fc1.0.weight"]), M, H, C, 1, addend=dx2, bn=_link_in(S) or False)
    _wgrad(dr4, x1, ops.w2d(G["fc1.0.weight"]), M, H, C)
    _store_partial(S, part)
    return dx1


# ------------------------------------------------------------------------------------------------ Downsample
def downsample_forward(x: Tensor, P, S: Optional[dict], B: int, N: int, training: bool) -> Tensor:
    """Conv2d 3x3 s2 p1 on the (B, C, N, 1) map + BatchNorm (graph_encoder.py:44-50). N even (every GraFP stage): the conv is a
    GEMM over a zero-padded strided VIEW of x (csrc/gemm.hip nsid_downsample3_*): no im2col matrix is written or read, in
    either direction. Odd N keeps the materialising form."""
    M, C = x.shape
    Co = P["conv.0.weight"].shape[0]
    No = ops.ds_out_nodes(N)
    view = N % 2 == 0
    col = None if view else ops.im2col3_fwd(x, B, N, C)
    if fold_eval(training, S):   # packed weight with the eval-mode BatchNorm folded in: built once per checkpoint
        gamma, beta, rm, rv, _, _ = _bn(P, S, "conv.1.")
        wf, bf = ops.folded_conv_bn(lambda: ops.pack_ds_weight(P["conv.0.weight"]), P["conv.0.bias"], gamma, beta, rm, rv,
                                    source=P["conv.0.weight"])
        if view:
            return ops.downsample3_fwd(x, B, N, C, wf, bf, Co)[0]
        return ops.linear_fwd(col, wf, bf, B * No, Co, 3 * C)[0]
    prep = ops.DS_PREP.get(P["conv.0.weight"]) if (view and S is not None and ACT_DTYPE == torch.bfloat16) else None
    if prep is not None:
        # prepared once per step for both views (ops.DsPrep): packed weights already in bf16, one shared packed gradient buffer
        gamma, beta, rm, rv, nbt, _ = _bn(P, S, "conv.1.")
        r, stat = ops.downsample3_fwd(x, B, N, C, None, P["conv.0.bias"], Co, want_stat=training, w16=prep["wp16"])
        aff = bn_affine_from(stat, B * No, gamma, beta, rm, rv, nbt, training)
        out = ops.bn_apply(r, aff, ACT_NONE)
        S.update(col=None, x=x, wp=None, prep=prep, dwp_slot=ops.DS_PREP.take_slot(prep), r=r, aff=aff, B=B, N=N, C=C)
        if CHAIN is not None:
            CHAIN.produce(S, r, aff, ACT_NONE)
        return out
    wp = ops.pack_ds_weight(P["conv.0.weight"])
    if ACT_DTYPE == torch.bfloat16 and S is not None:
        # the packed weight lives until this block's backward: one bf16 conversion serves the forward and the backward-data
        # GEMM (an unregistered weight is converted at every use)
        ops.SHADOWS.register(wp, ops.f32_to_bf16(wp), owner=wp, fresh=True)
    if view:
        gamma, beta, rm, rv, nbt, _ = _bn(P, S, "conv.1.")
        r, stat = ops.downsample3_fwd(x, B, N, C, wp, P["conv.0.bias"], Co, want_stat=training)
        aff = bn_affine_from(stat, B * No, gamma, beta, rm, rv, nbt, training)
    else:
        r, aff = conv_bn(col, B * No, 3 * C, Co, wp, P["conv.0.bias"], _bn(P, S, "conv.1."), training)
    out = ops.bn_apply(r, aff, ACT_NONE)
    if S is not None:
        S.update(col=col, x=x if view else None, wp=wp, r=r, aff=aff, B=B, N=N, C=C)
        if CHAIN is not None:
            CHAIN.produce(S, r, aff, ACT_NONE)      # (its own input gradient is written by two GEMMs / col2im: nothing to fuse)
    return out


def downsample_backward(dout: Tensor, P, S, G) -> Tensor:
    col, x, wp, r, aff, B, N, C = (S[k_] for k_ in ("col", "x", "wp", "r", "aff", "B", "N", "C"))
    Mo, Co = r.shape
    dr = ops.bn_backward(dout, r, aff, ACT_NONE, G["conv.1.weight"], G["conv.1.bias"], partial=_link_partial(S))
    _bias_grad_before_bn(dr, G["conv.0.bias"])
    prep = S.get("prep")
    if prep is not None:
        dwp = S.get("dwp_slot")                       # zeroed by the step's prepack launch; None: both slots were taken
        if dwp is None:
            dwp = ops.zeros(prep["dwp"].shape[1:], prep["dwp"].device)
        if _deferring() and DEFER_DS and dr.dtype == torch.bfloat16:
            # both views add into the layer's FIRST packed buffer (one problem with two row segments), unpacked once after the grouped launch
            if S.get("dwp_slot") is not None:
                dwp = prep["dwp"][0]
            DEFERRED.add(dr, x, dwp, Mo, Co, 3 * C, 1, None, None, ACT_NONE, ("ds", B, N, C, G["conv.0.weight"]))
        else:
            ops.downsample3_bwd_weight(dr, x, dwp, B, N, C, Co)
            ops.unpack_ds_wgrad(dwp, G["conv.0.weight"])
        return ops.downsample3_bwd_data(dr, None, None, B, N, C, Co, w16=(prep["wp16"], prep["wb16"]))
    dwp = ops.zeros(wp.shape, wp.device)
    if col is None:                                  # strided-view form: x is the operand
        ops.downsample3_bwd_weight(dr, x, dwp, B, N, C, Co)
        ops.unpack_ds_wgrad(dwp, G["conv.0.weight"])
        return ops.downsample3_bwd_data(dr, wp, ops.pack_ds_weight_bwd(P["conv.0.weight"]), B, N, C, Co)
    ops.linear_bwd_weight(dr, col, dwp, Mo, Co, 3 * C)
    ops.unpack_ds_wgrad(dwp, G["conv.0.weight"])
    dcol = ops.linear_bwd_data(dr, wp, Mo, Co, 3 * C)
    return ops.im2col3_bwd(dcol, B, N, C)


# ------------------------------------------------------------------------------------------------ proj + node mean
HEAD_FC1_KSPLIT = 1     # reduction split of the projector's first linear: measured 2 / 4 against 1: 7.787 / 7.787 against 7.773 ms (x3): off
HEAD_KSPLIT = True      # split the 512-deep reduction of the B-row projection GEMM (one-box A/B: docs/experiments.md, round 5)


def proj_mean_forward(x: Tensor, P, S: Optional[dict], B: int, N: int) -> Tensor:
    """mean over nodes commutes with the 1x1 projection: h = W mean_n(x) + b (32x fewer FLOPs than proj-then-mean)"""
    M, C = x.shape
    E = P["weight"].shape[0]
    xm = ops.node_mean_fwd(x, B, N, C)
    # B rows: a handful of tiles with a 512-deep reduction — split it (fp32 atomics into the zeroed output), as the projector's fc2
    # (training only: S is None in inference, whose embeddings then do not depend on atomics order or on the launch size -- ADVICE r5)
    h, _ = ops.linear_fwd(xm, ops.w2d(P["weight"]), P["bias"], B, E, C,
                          ksplit=4 if (HEAD_KSPLIT and S is not None and C >= 512 and B <= 512) else 1)
    if S is not None:
        S.update(xm=xm, B=B, N=N, C=C, xdtype=x.dtype)
    return h


def proj_mean_backward(dh: Tensor, P, S, G) -> Tensor:
    xm, B, N, C = S["xm"], S["B"], S["N"], S["C"]
    E = dh.shape[1]
    _wgrad(dh, xm, ops.w2d(G["weight"]), B, E, C)
    _bias_grad(dh, G["bias"])
    dxm = ops.linear_bwd_data(dh, ops.w2d(P["weight"]), B, E, C)
    return ops.node_mean_bwd(dxm, B, N, C, S["xdtype"])


# ------------------------------------------------------------------------------------------------ projector
def projector_forward(h: Tensor, P, S: Optional[dict], eps: float = 1e-10) -> Tensor:
    B, Hin = h.shape
    Hid, D = P["0.weight"].shape[0], P["2.weight"].shape[0]
    a1, _ = ops.linear_fwd(h, P["0.weight"], P["0.bias"], B, Hid, Hin, act_out=ACT_ELU,
                           ksplit=HEAD_FC1_KSPLIT if (Hin >= 1024 and B <= 512 and h.dtype == torch.float32) else 1)
    ks = 8 if (Hid >= 2048 and S is not None) else 1        # split reductions (fp32 atomics) in training only
    p, _ = ops.linear_fwd(a1, P["2.weight"], P["2.bias"], B, D, Hid, ksplit=ks)
    z, norm = ops.l2norm_fwd(p, eps)
    if S is not None:
        S.update(h=h, a1=a1, z=z, norm=norm, eps=eps)
    return z


def projector_backward(dz: Tensor, P, S, G) -> Tensor:
    h, a1, z, norm, eps = S["h"], S["a1"], S["z"], S["norm"], S["eps"]
    B, Hin = h.shape
    Hid, D = a1.shape[1], z.shape[1]
    dp = ops.l2norm_bwd(dz, z, norm, eps)
    _wgrad(dp, a1, G["2.weight"], B, D, Hid)
    _bias_grad(dp, G["2.bias"])
    da1 = ops.linear_bwd_data(dp, P["2.weight"], B, D, Hid)
    dpre = ops.elu_bwd(da1, a1)
    _wgrad(dpre, h, G["0.weight"], B, Hid, Hin)
    _bias_grad(dpre, G["0.bias"])
    return ops.linear_bwd_data(dpre, P["0.weight"], B, Hid, Hin)


# ------------------------------------------------------------------------------------------------ peak extractor
def patchify_forward(spec: Tensor, P, S: Optional[dict], pb: int, pf: int) -> Tensor:
    out, minmax = ops.peak_patchify_fwd(spec, P["convs.0.weight"], P["convs.0.bias"], pb, pf, ACT_DTYPE)
    if S is not None:
        S.update(spec=spec, minmax=minmax, out=out, pb=pb, pf=pf)
    return out


DEFER_PATCHIFY = 1      # the peak extractor's backward (parameter gradients only; the last launches of a view's chain, fully exposed: 27 us)
#                         joins the deferred phase's second lane


def patchify_backward(dout: Tensor, P, S, G) -> None:
    args = (S["spec"], S["minmax"], S["out"], dout, S["pb"], S["pf"], G["convs.0.weight"], G["convs.0.bias"])
    if _deferring() and DEFER_PATCHIFY:
        DEFERRED.add_call(lambda: ops.peak_patchify_bwd(*args), args[:4])
    else:
        ops.peak_patchify_bwd(*args)


# ================================================================================================ autograd
class _BlockFn(torch.autograd.Function):
    """Generic wrapper: forward(fwd, bwd, names, buffers, meta, x, *params) with P assembled from names."""

    @staticmethod
    def forward(ctx, fwd, bwd, names, buffers, meta, grad_on, x, *params):
        P = dict(zip(names, params))
        P.update(buffers)
        # grad_on: torch.is_grad_enabled() at the call site (inside Function.forward grad mode is always off, and
        # needs_input_grad stays True for parameters under torch.no_grad(): without it inference kept a saved-tensor dict)
        need = grad_on and any(ctx.needs_input_grad)
        S = {} if need else None
        out = fwd(x.contiguous(), P, S, *meta)
        ctx.S, ctx.P, ctx.bwd, ctx.names = S, P, bwd, names
        ctx.x_needs = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        P, S, names = ctx.P, ctx.S, ctx.names
        direct = DIRECT_GRADS and all(P[n].grad is not None and P[n].grad.is_contiguous() for n in names)
        G = {n: (P[n].grad if direct else ops.zeros(P[n].shape, P[n].device)) for n in names}
        global _IN_DIRECT_BACKWARD
        n_def = len(DEFERRED.items)
        _IN_DIRECT_BACKWARD = direct
        try:
            dx = ctx.bwd(dout.contiguous(), P, S, G)
        finally:
            _IN_DIRECT_BACKWARD = False
        ctx.S = None
        if direct and GRAD_READY_HOOK is not None:
            if len(DEFERRED.items) > n_def:      # this block's gradients are complete only after the deferred phase
                DEFERRED.note_hook([P[n] for n in names], n_def)
            else:
                GRAD_READY_HOOK([P[n] for n in names])
        head = (None, None, None, None, None, None, dx if ctx.x_needs else None)
        return head + tuple(None if direct else G[n] for n in names)


def run_block(fwd, bwd, module_params: Dict[str, Tensor], module_buffers: Dict[str, Tensor], x: Tensor, *meta):
    names = tuple(module_params.keys())
    return _BlockFn.apply(fwd, bwd, names, module_buffers, meta, torch.is_grad_enabled(), x, *module_params.values())


class _ToRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):           # (B, C, N[,1]) fp32 -> (B*N, C) in the activation storage type
        ctx.shape = x.shape
        return ops.bcn_to_rows(x.contiguous(), ACT_DTYPE)

    @staticmethod
    def backward(ctx, d):
        B, N = ctx.shape[0], ctx.shape[2]
        return ops.rows_to_bcn(d.contiguous(), B, N).reshape(ctx.shape)


class _FromRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, B, N):  # (B*N, C) -> (B, C, N) fp32
        ctx.rows_dtype = rows.dtype
        return ops.rows_to_bcn(rows.contiguous(), B, N)

    @staticmethod
    def backward(ctx, d):
        return ops.bcn_to_rows(d.contiguous(), ctx.rows_dtype), None, None


def to_rows(x: Tensor) -> Tensor:
    return _ToRows.apply(x)


def from_rows(rows: Tensor, B: int, N: int) -> Tensor:
    return _FromRows.apply(rows, B, N)


class _NtxentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_i, z_j, tau, grad_on=True):
        need = grad_on and any(ctx.needs_input_grad)
        loss, dzi, dzj = ops.ntxent_fwd_bwd(z_i.contiguous(), z_j.contiguous(), tau, want_grad=need)
        ctx.save_for_backward(dzi, dzj) if need else None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dzi, dzj = ctx.saved_tensors
        return ops.scale_f32(dzi, g), ops.scale_f32(dzj, g), None, None


def ntxent(z_i: Tensor, z_j: Tensor, tau: float) -> Tensor:
    return _NtxentFn.apply(z_i, z_j, tau, torch.is_grad_enabled())
