"""Tensor-level wrappers over the C ABI (include/nsid.h): torch supplies device memory and the stream, nothing else.

Every function enqueues HIP kernels from libnsid_hip.so on torch's current stream; none falls back to ATen."""
import os
from typing import Optional, Tuple

import torch

from ._lib import call, get_tuning, launch_counters, lib, reset_tuning, set_tuning  # noqa: F401  (tuning, counters: re-exported)

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3
ROW_TILE = 128
BN_MOMENTUM = 0.1
BN_EPS = 1e-5


class KernelProfile:
    """Opt-in per-launch timing of the GEMM family with HIP events recorded on the launch stream (bench.py roofline).
    Keys follow the kernel template instantiation that the C side dispatches to (csrc/gemm.hip)."""

    def __init__(self):
        self.records = []          # (kernel, flops, algorithmic bytes, start event, end event, shape)
        self.bracket_ms = self._empty_bracket_ms()

    @staticmethod
    def plug(ms: float = 120.0) -> None:
        """Keep the GPU busy for ~`ms` so that the host runs AHEAD of it: an eager step is host-bound (~7 us of Python / ctypes
        per launch), and an event pair around a launch on an idle GPU also measures the wait for the launch packet
        (a tiny kernel reads 12.4 us, a 19.4 us GEMM 27 us). Behind a plug the queue is full, kernels run back to back and an
        event pair brackets the kernel alone (the same GEMM reads 21.2 us: rocprofv3 says 20.6 us). Measurement aid only:
        never inside a timed region."""
        torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))

    @staticmethod
    def _empty_bracket_ms(n: int = 64) -> float:
        """median elapsed time of an EMPTY event bracket on the current stream: two timed events cost a marker each, and
        that cost sits inside every per-launch measurement of an eager step; it is subtracted in summary()/by_shape()"""
        torch.cuda.synchronize()
        KernelProfile.plug(2.0)
        evs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2]

    def _ms(self, e0, e1) -> float:
        return max(e0.elapsed_time(e1) - self.bracket_ms, 1e-4)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1, shape in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += self._ms(e0, e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out

    def by_shape(self):
        """per (kernel, M, Nout, K, groups): launches, average us, TFLOP/s, algorithmic GB/s"""
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1, shape in self.records:
            shape = tuple(shape)[:4] + (0,) * (4 - min(4, len(shape)))       # records without a GEMM shape: zeros
            d = out.setdefault((name,) + shape, [0, 0.0, flops, nbytes])
            d[0] += 1
            d[1] += self._ms(e0, e1)
        rows = []
        for key, (n, ms, flops, nbytes) in out.items():
            us = 1e3 * ms / n
            rows.append({"kernel": key[0], "M": key[1], "Nout": key[2], "K": key[3], "groups": key[4], "launches": n,
                         "avg_us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
                         "alg_GBps": round(nbytes / us / 1e3, 0), "total_ms": round(ms, 3)})
        return sorted(rows, key=lambda r: -r["total_ms"])


PROFILE: Optional[KernelProfile] = None


def _timed(name, flops, nbytes, fn, shape=()):
    """name: a string, or a callable evaluated AFTER the launch (a launcher that picks its kernel in C reports it then)"""
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    PROFILE.records.append((name() if callable(name) else name, flops, nbytes, e0, e1, shape))


def _cnt(key: str) -> int:
    """a launch counter of the library (profiling only: which kernel a launcher that decides in C has just taken)"""
    return int(lib.nsid_debug_counter(key.encode())) if PROFILE is not None else 0


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    """parameters / statistics / head tensors: contiguous fp32 on the GPU"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("neuralsampleid_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")


F32, BF16 = 0, 1


def _tk(name, nbytes, fn, shape=()):
    """time a launch that moves bytes and does no matrix work (BatchNorm family, streaming passes, optimiser): bench.py's kernel table"""
    return _timed(name, 0.0, float(nbytes), fn, shape)


def _act(*ts) -> int:
    """activation tensors (features, raw conv outputs, their gradients): contiguous fp32 or bf16, all the same type;
    returns the C ABI dtype code"""
    code = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("neuralsampleid_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
        if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
            raise RuntimeError(f"expected contiguous float32/bfloat16, got {t.dtype} contiguous={t.is_contiguous()}")
        c = BF16 if t.dtype == torch.bfloat16 else F32
        if code is not None and c != code:
            raise RuntimeError("activation tensors of one call must share a storage type")
        code = c
    return code


GEMM_FP32, GEMM_BF16 = 0, 1


def set_gemm_precision(mode: str) -> None:
    """'fp32' (default; exact fp32 MFMA, the strict-parity path) or 'bf16' (operands rounded to bf16 as they are
    staged into LDS; fp32 storage in HBM, fp32 accumulation; 16x the matrix rate — BASELINE config 2's arithmetic)."""
    call("nsid_set_gemm_precision", {"fp32": GEMM_FP32, "bf16": GEMM_BF16}[mode])


def get_gemm_precision() -> str:
    return "bf16" if lib.nsid_get_gemm_precision() == GEMM_BF16 else "fp32"


def row_tiles(M: int) -> int:
    return (M + ROW_TILE - 1) // ROW_TILE


def w2d(w: torch.Tensor) -> torch.Tensor:
    """conv weight (Cout, Cin/g, 1, 1) or linear weight (Cout, Cin) viewed as a matrix (no copy)"""
    return w.reshape(w.shape[0], -1)


# ------------------------------------------------------------------------------------------------ state epoch
# Our kernels write parameters (nsid_adam_step) and BatchNorm running statistics (nsid_bn_finalize, nsid_bn_running_update)
# through raw pointers, and a hipGraph replay runs no Python at all: torch's `_version` counters do not see those writes.
# Every host-side entry that enqueues such a write bumps an epoch counter: WEIGHT_EPOCH for parameters
# (FusedClipAdam.step, GraphedTrainStep.__call__), STATS_EPOCH for running statistics (bn_finalize, bn_running_update,
# GraphedTrainStep.__call__). The caches below carry them in their keys — bf16 weight shadows: WEIGHT_EPOCH; eval-mode
# BatchNorm affines and folded conv+BN weights: both — so an eval after training steps never reuses constants of older
# weights or statistics.
WEIGHT_EPOCH = 0
STATS_EPOCH = 0


def bump_state_epoch(weights: bool = True, stats: bool = True) -> None:
    global WEIGHT_EPOCH, STATS_EPOCH
    if weights:
        WEIGHT_EPOCH += 1
    if stats:
        STATS_EPOCH += 1


# ------------------------------------------------------------------------------------------------ bf16 weight shadows
class WeightShadows:
    """bf16 copies of weight matrices for the bf16-storage GEMMs (forward / backward-data right operand): half the
    L2->LDS operand bytes and no convert pass in the kernel. Values are the RNE rounding the kernel applies to fp32
    weights anyway, so results are identical.

    Entries are keyed by the fp32 tensor's data pointer: [shadow, (version, WEIGHT_EPOCH)]. `FusedClipAdam` registers views
    of ONE flat shadow buffer, refreshes all of it at the start of every step and marks its entries fresh (`mark_fresh`);
    its update kernel bumps WEIGHT_EPOCH instead of torch's version counters, so a forward that runs after the last
    `step()` (eval, fingerprint extraction) re-converts the weights it reads. Torch-side in-place changes
    (load_state_dict, torch.optim) are caught by the version half of the key. Weights that are not registered are
    converted on the fly (one small launch per GEMM)."""

    def __init__(self):
        self.entries = {}

    def register(self, w: torch.Tensor, shadow: torch.Tensor, owner: torch.Tensor, fresh: bool = False) -> None:
        """`owner` keeps the fp32 memory alive (the optimiser's flat buffer): while it lives the address cannot be
        reused by another tensor; once it is gone the entry is dropped on its next lookup. fresh: `shadow` already holds
        the conversion of the current contents of w"""
        import weakref
        self.entries[w.data_ptr()] = [shadow, (w._version, WEIGHT_EPOCH) if fresh else None, w.numel(), weakref.ref(owner)]

    def mark_fresh(self, w: torch.Tensor) -> None:
        """the registered shadow of `w` has just been refreshed by its owner (FusedClipAdam.sync_shadow)"""
        e = self.entries.get(w.data_ptr())
        if e is not None:
            e[1] = (w._version, WEIGHT_EPOCH)

    def clear(self) -> None:
        self.entries.clear()

    def purge(self) -> int:
        """drop the entries whose fp32 memory is gone (their shadows would otherwise stay alive with the registry)"""
        dead = [k for k, e in self.entries.items() if e[3]() is None]
        for k in dead:
            del self.entries[k]
        return len(dead)

    def operand(self, w: torch.Tensor) -> torch.Tensor:
        e = self.entries.get(w.data_ptr())
        if e is not None and e[3]() is None:
            del self.entries[w.data_ptr()]
            e = None
        if e is None or e[2] != w.numel():
            return f32_to_bf16(w)
        key = (w._version, WEIGHT_EPOCH)
        if e[1] != key:
            f32_to_bf16(w, e[0])
            e[1] = key
        return e[0]


SHADOWS = WeightShadows()


def register_weight_shadows(module) -> int:
    """Give every not-yet-registered GEMM weight of `module` a persistent bf16 shadow (inference: no optimiser keeps one).
    Shadows follow torch-side weight changes through the version check. Returns the number of new shadows."""
    n = 0
    for p in module.parameters():
        if p.is_cuda and p.dim() >= 2 and p.numel() % 8 == 0 and p.data_ptr() not in SHADOWS.entries:
            SHADOWS.register(p.data, torch.empty(p.numel(), device=p.device, dtype=torch.bfloat16), owner=p)
            n += 1
    return n


def f32_to_bf16(src: torch.Tensor, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(src)
    n = src.numel()
    if dst is None:
        dst = torch.empty(((n + 7) // 8 * 8,), device=src.device, dtype=torch.bfloat16)
    if n % 8:            # tiny odd-sized matrices: let torch round (same RNE)
        dst[:n].copy_(src.reshape(-1))
        return dst
    _tk("f32_to_bf16_kernel", 6.0 * n, lambda: call("nsid_f32_to_bf16", _p(src), _p(dst), n, _stream()), (n, 1, 0, 1))
    return dst


def _weight(w: torch.Tensor, dt: int, K: int):
    """(pointer-holder tensor, C ABI dtype code) of a GEMM's weight operand"""
    if dt == BF16 and K % 8 == 0 and w.numel() % 8 == 0:
        return SHADOWS.operand(w), BF16
    return w, F32


# ------------------------------------------------------------------------------------------------ linear
def linear_fwd(x, w, bias, M, Nout, K, groups=1, in_scale=None, in_shift=None, act_in=ACT_NONE, act_out=ACT_NONE,
               want_stat=False, ksplit=1, out=None, addend=None, in_aff=None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """out = f(x) w^T + bias (+ addend, same layout and storage as out: the residual stream, nsid_linear_fwd_res).
    in_aff: the producer's BNAffine instead of (in_scale, in_shift)."""
    if in_aff is not None:
        in_scale, in_shift = in_aff.scale, in_aff.shift
    _chk(w, bias, in_scale, in_shift)
    dt = _act(x, out, addend)
    if addend is not None and (want_stat or ksplit != 1 or act_out != ACT_NONE):
        raise ValueError("the residual addend rides with a plain epilogue (no statistics, no split-K, no output activation)")
    ldx = x.shape[-1]
    if out is None:
        out = torch.empty((M, groups * Nout), device=x.device, dtype=x.dtype)
        if ksplit > 1:
            fill_zero(out)
    stat = torch.empty((2, row_tiles(M), groups * Nout), device=x.device, dtype=torch.float32) if want_stat else None
    half = dt == BF16 or lib.nsid_get_gemm_precision() == GEMM_BF16
    narrow = (Nout <= 64 or (Nout <= 128 and K <= 256 and in_scale is None)) if half else Nout <= 64
    if half and row_tiles(M) * ((Nout + 127) // 128) * groups < 128:
        narrow = True
    esz = x.element_size()
    wop, wdt = _weight(w, dt, K)
    name_ = "gemm_kernel<128,%d,true,true>" % (64 if narrow else 128)
    n256 = lib.nsid_gemm_g256_launches() if PROFILE is not None else 0
    nws = _cnt("ws_fwd")
    # csrc/gemm.hip decides between the kernel families; the launch counters say which one ran
    name = lambda: ("ws_fwd_kernel" if _cnt("ws_fwd") > nws else
                    ("gemm256_fwd_kernel" if lib.nsid_gemm_g256_launches() > n256 else name_))
    if in_scale is None and act_in != ACT_NONE:
        # activation on load without an affine: only ReLU on bf16 operands, bf16 weights and whole tiles (csrc/gemm.hip ARELU)
        bn_ = 64 if narrow else 128
        bm_ = 128
        if not (act_in == ACT_RELU and dt == BF16 and wdt == BF16 and M % bm_ == 0 and Nout % bn_ == 0 and K % 64 == 0
                and ksplit == 1):
            raise ValueError("activation-on-load without an affine needs ReLU, bf16 storage, bf16 weights and whole tiles")
    if addend is not None:
        _timed(name, 2.0 * M * Nout * K * groups,
               groups * (esz * M * K + float(wop.element_size()) * Nout * K + 2 * esz * M * Nout), lambda: call(
            "nsid_linear_fwd_res", _p(x), ldx, _p(wop), wdt, _p(bias), _p(addend), addend.shape[-1], _p(out),
            out.shape[-1], M, Nout, K, groups, _p(in_scale), _p(in_shift), act_in, dt, _stream()), (M, Nout, K, groups))
        return out, None
    _timed(name, 2.0 * M * Nout * K * groups,
           groups * (esz * M * K + float(wop.element_size()) * Nout * K + esz * M * Nout), lambda: call(
        "nsid_linear_fwd", _p(x), ldx, _p(wop), wdt, _p(bias), _p(out), out.shape[-1], M, Nout, K, groups, _p(in_scale),
        _p(in_shift), act_in, act_out, _p(stat), ksplit, dt, _stream()), (M, Nout, K, groups))
    return out, stat


FUSE_EVAL_FFN = True          # eval-mode FFN (BatchNorms folded) as one launch for the C = 64 / 128 stages (csrc/ffn_fused.hip)


def ffn_fused_fwd(x, w1f, b1f, w2f, b2f, M, C, H):
    """out = x + w2f relu(w1f x + b1f) + b2f in one launch (folded eval-mode FFN); None when the shape is outside the fused form"""
    _chk(w1f, b1f, w2f, b2f)
    if _act(x) != BF16 or x.shape != (M, C) or not FUSE_EVAL_FFN:
        return None
    w1, d1 = _weight(w1f, BF16, C)
    w2, d2 = _weight(w2f, BF16, H)
    if d1 != BF16 or d2 != BF16:
        return None
    out = torch.empty_like(x)
    rc = [0]

    def launch():
        rc[0] = lib.nsid_ffn_fused_fwd(_p(x), _p(w1), _p(b1f), _p(w2), _p(b2f), _p(out), M, C, H, _stream())
    _timed("ffn_fused_kernel", 4.0 * M * C * H, 2.0 * x.element_size() * M * C + 4.0 * C * H, launch, (M, C, H, 1))
    if rc[0] == 1:
        if PROFILE is not None:
            PROFILE.records.pop()
        return None
    if rc[0] != 0:
        raise RuntimeError(f"nsid_ffn_fused_fwd failed: {rc[0]}")
    return out


FUSE_EVAL_MRCONV = True       # eval-mode max-relative aggregation + grouped conv as one launch (csrc/mrconv_fused.hip)


def mrconv_fused_fwd(y, idx, B, N, C, wf, bf):
    """relu(grouped conv([y, max-relative(y)]) + b) with the BatchNorm folded into (wf, bf), one workgroup per clip; the interleaved
    (B*N, 2C) tensor is never formed. None when the shape is outside the fused form."""
    _chk(wf, bf)
    if not FUSE_EVAL_MRCONV or _act(y) != BF16 or y.shape != (B * N, C) or idx.dtype != torch.int32 or not idx.is_contiguous():
        return None
    w, dw = _weight(wf, BF16, C // 2)
    if dw != BF16:
        return None
    k = idx.shape[-1]
    out = torch.empty((B * N, 2 * C), device=y.device, dtype=y.dtype)
    rc = [0]

    def launch():
        rc[0] = lib.nsid_mrconv_fused_fwd(_p(y), _p(idx), B, N, C, k, _p(w), _p(bf), _p(out), _stream())
    _timed("mrconv_fused_kernel", 2.0 * B * N * (C // 2) * (C // 2) * 4, float(B) * N * (3 * C * y.element_size() + k * 4), launch,
           (B * N, C // 2, C // 2, 4))
    if rc[0] == 1:
        if PROFILE is not None:
            PROFILE.records.pop()
        return None
    if rc[0] != 0:
        raise RuntimeError(f"nsid_mrconv_fused_fwd failed: {rc[0]}")
    return out


FUSE_BN_BWD_REDUCE = True     # backward-data GEMMs emit the next BatchNorm-backward's column sums (bf16 storage only)


def linear_bwd_data(dout, w, M, Nout, K, groups=1, addend=None, out=None, bn=None):
    """din = addend + dout @ w.  bn = (r, BNAffine, act): din is dL/dy of the layer y = act(BN(r)); then the call returns
    (din, partial) with partial[2][tiles][groups*K] = that layer's backward column sums (see bn_backward(partial=...))."""
    _chk(w)
    dt = _act(dout, addend, out)
    if out is None:
        out = torch.empty((M, groups * K), device=dout.device, dtype=dout.dtype)
    half = dt == BF16 or lib.nsid_get_gemm_precision() == GEMM_BF16
    narrow = (K <= 64 or (K <= 128 and Nout <= 256)) if half else K <= 64
    if half and row_tiles(M) * ((K + 127) // 128) * groups < 128:
        narrow = True
    name_ = "gemm_kernel<128,%d,true,false>" % (64 if narrow else 128)
    nws = _cnt("ws_bwd_data")
    name = lambda: "ws_bwd_kernel" if _cnt("ws_bwd_data") > nws else name_
    esz = dout.element_size()
    wop, wdt = _weight(w, dt, K)
    want_pair = bn is not None            # bn=False: "return a pair, nothing to fuse"
    bn = bn or None
    fuse = bn is not None and FUSE_BN_BWD_REDUCE and dt == BF16 and out.shape[-1] == groups * K and (groups * K) % 8 == 0
    if fuse and tuple(bn[0].shape) != (M, groups * K):
        fuse = False
    nbytes = groups * (esz * M * Nout + float(wop.element_size()) * Nout * K
                       + esz * M * K * ((2 if addend is not None else 1) + (1 if fuse else 0)))
    if fuse:
        r, aff, act = bn
        _act(r, out)
        partial = torch.empty((2, row_tiles(M), groups * K), device=dout.device, dtype=torch.float32)
        _timed(name, 2.0 * M * Nout * K * groups, nbytes, lambda: call(
            "nsid_linear_bwd_data_bn", _p(dout), dout.shape[-1], _p(wop), wdt, _p(addend),
            0 if addend is None else addend.shape[-1], _p(out), out.shape[-1], M, Nout, K, groups, dt, _p(r),
            _p(aff.scale), _p(aff.shift), _p(aff.mean), _p(aff.invstd), act, _p(partial), _stream()),
            (M, Nout, K, groups))
        return out, partial
    _timed(name, 2.0 * M * Nout * K * groups, nbytes, lambda: call(
        "nsid_linear_bwd_data", _p(dout), dout.shape[-1], _p(wop), wdt, _p(addend),
        0 if addend is None else addend.shape[-1], _p(out), out.shape[-1], M, Nout, K, groups, dt, _stream()),
        (M, Nout, K, groups))
    return (out, None) if want_pair else out


# BatchNorm-backward apply evaluated on the operand load of the backward-data GEMM (bf16 storage): bit mask over the call sites
# (1: FFN hidden layer, 2: grouped conv, 4: Grapher fc1); True = all, False / 0 = none
FUSE_BN_BWD_APPLY = 4        # measured in the two-stream step (round 3, one-box A/B x2): none 8.09 / 8.06 ms, FFN 8.35 / 8.35, grouped 8.14 / 8.13, fc1 8.08 / 8.09, all 8.33 / 8.32
SITE_FFN, SITE_GCONV, SITE_FC1, SITE_FC2, SITE_FFN2 = 1, 2, 4, 8, 16
# the backward-data shapes (Nout, K, groups) of csrc/wsgemm.hip: there the fused operand load costs one evaluation per ROW (a workgroup owns
# all output columns), not one per column tile, so every call site takes it (tuning key ws_gemm bit 2); elsewhere FUSE_BN_BWD_APPLY decides
WS_BWD_SHAPES = {(64, 64, 1), (32, 32, 4), (64, 128, 1), (256, 64, 1), (64, 256, 1), (128, 128, 1), (64, 64, 4), (128, 256, 1),
                 (128, 128, 4)}


def bn_backward_linear_bwd_data(dy, r, aff: "BNAffine", act, dgamma, dbeta, partial, w, M, Nout, K, groups=1, addend=None,
                                bn=None, site=SITE_FFN, inplace=True):
    """BatchNorm(+act) backward of a conv+BN layer FOLLOWED by that conv's backward-data GEMM, the two fused where the shape allows:
        dr  = BN-backward(dy, r)        (dy = dL/d act(BN(r)); dgamma / dbeta accumulated)
        din = addend + dr @ w           (+ the column sums for the next BatchNorm backward when bn=(r', aff', act'))
    Returns (dr, din, partial_out). Fused form (csrc/gemm.hip ABN): the GEMM evaluates dr on its operand load from dy and r and
    writes it once for the weight gradient — no bn_bwd_apply pass. `partial` = this layer's column sums (from the GEMM that wrote dy,
    or None: a reduce launch). Falls back to bn_backward + linear_bwd_data (same results up to one fp32 rounding before the bf16
    store) for fp32 storage and shapes outside the fused form. inplace: the unfused fallback may overwrite dy with dr (False where the
    caller still needs dy: the residual-stream gradients dx1 / dx2 are also the addends of a later GEMM)."""
    dt = _act(dy, r, addend)
    C = r.shape[1]
    tiles = row_tiles(M)
    s = _stream()
    wop, wdt = _weight(w, dt, K)
    enabled = FUSE_BN_BWD_APPLY is True or (FUSE_BN_BWD_APPLY and (int(FUSE_BN_BWD_APPLY) & site))
    ws_form = (dt == BF16 and wdt == BF16 and M % 128 == 0 and (Nout, K, groups) in WS_BWD_SHAPES and (get_tuning("ws_gemm") & 4) != 0)
    enabled = enabled or ws_form
    fusable = (enabled and dt == BF16 and wdt == BF16 and tuple(r.shape) == (M, groups * Nout) and
               tuple(dy.shape) == (M, groups * Nout) and M % 128 == 0 and
               (ws_form or (Nout % 64 == 0 and K % 64 == 0 and (K <= 64 or K % 128 == 0))))
    if not fusable:
        dr = bn_backward(dy, r, aff, act, dgamma, dbeta, inplace=inplace, partial=partial)
        return (dr,) + linear_bwd_data(dr, w, M, Nout, K, groups, addend=addend, bn=bn if bn is not None else False)
    if partial is None:
        partial = torch.empty((2, tiles, C), device=r.device, dtype=torch.float32)
        part_ = partial
        _tk("col_reduce_kernel", 2.0 * r.element_size() * M * C, lambda: call(
            "nsid_bn_bwd_reduce", _p(dy), _p(r), M, C, _p(aff.scale), _p(aff.shift), _p(aff.mean), _p(aff.invstd), act, _p(part_),
            dt, s), (M, C, 0, 1))
    ptiles = partial.shape[1]                  # rows of THIS layer's partial sums (row tiles, or one per clip from mr_aggregate_bwd)
    coef = torch.empty((6, C), device=r.device, dtype=torch.float32)        # [0:2] = (c0, c1) for the unfused apply, [2:6] = coef4
    if not (DIAG_SKIP_FINALIZE & 2):
        _tk("bn_bwd_finalize_kernel", 8.0 * ptiles * C + 40.0 * C, lambda: call(
            "nsid_bn_bwd_finalize_fused", _p(partial), ptiles, C, M, _p(dgamma), _p(dbeta), _p(coef), _p(aff.scale), _p(aff.shift),
            _p(aff.mean), _p(aff.invstd), _p(coef[2]), s), (M, C, 0, 1))
    dr = torch.empty_like(dy)
    din = torch.empty((M, groups * K), device=dy.device, dtype=dy.dtype)
    want_pair = bn is not None
    bn = bn or None
    fuse_sums = bn is not None and FUSE_BN_BWD_REDUCE and (groups * K) % 8 == 0 and tuple(bn[0].shape) == (M, groups * K)
    part_out = torch.empty((2, tiles, groups * K), device=dy.device, dtype=torch.float32) if fuse_sums else None
    br, baff, bact = bn if fuse_sums else (None, None, ACT_NONE)
    esz = dy.element_size()
    nbytes = groups * (3 * esz * M * Nout + float(wop.element_size()) * Nout * K
                       + esz * M * K * ((2 if addend is not None else 1) + (1 if fuse_sums else 0)))
    rc = [0]

    def launch():
        rc[0] = lib.nsid_linear_bwd_data_bnapply(
            _p(dy), _p(r), _p(coef[2]), act, _p(dr), _p(wop), wdt, _p(addend), 0 if addend is None else addend.shape[-1], _p(din),
            din.shape[-1], M, Nout, K, groups, dt, _p(br), _p(baff.scale) if baff else None, _p(baff.shift) if baff else None,
            _p(baff.mean) if baff else None, _p(baff.invstd) if baff else None, bact, _p(part_out), s)
    _timed("ws_bwd_kernel +bn_apply_load" if ws_form else "gemm_kernel<128,%d,true,false> +bn_apply_load" % (64 if K <= 64 else 128),
           2.0 * M * Nout * K * groups, nbytes, launch, (M, Nout, K, groups))
    if rc[0] == 1:             # outside the fused form after all: apply with the coefficients already computed, then the plain GEMM
        if PROFILE is not None:
            PROFILE.records.pop()          # nothing was launched under that name
        call("nsid_bn_bwd_apply", _p(dy), _p(r), M, C, _p(aff.scale), _p(aff.shift), _p(aff.mean), _p(aff.invstd), act, _p(coef),
             _p(dr), dt, s)
        res = linear_bwd_data(dr, w, M, Nout, K, groups, addend=addend, out=din, bn=bn if bn is not None else False)
        return (dr,) + res
    if rc[0] != 0:
        raise RuntimeError(f"nsid_linear_bwd_data_bnapply failed: {rc[0]}")
    return dr, din, part_out


def linear_bwd_weight(dout, x, dw, M, Nout, K, groups=1, in_scale=None, in_shift=None, act_in=ACT_NONE) -> None:
    """dw += dout^T f(x)"""
    _chk(dw, in_scale, in_shift)
    dt = _act(dout, x)
    t128 = ((Nout + 127) // 128) * ((K + 127) // 128) * groups
    half = dt == BF16 or lib.nsid_get_gemm_precision() == GEMM_BF16
    small = half or Nout <= 64 or K <= 64 or t128 * ((M + 511) // 512) < 256
    name = "gemm_kernel<%s,false,false>" % ("64,64" if small else "128,128")
    if dt == BF16 and Nout % 128 == 0 and K % 64 == 0 and M % 1024 == 0:          # csrc/gemm.hip: 128x64 tiles
        tiles_r = (Nout // 128) * (K // 64) * groups
        if tiles_r * min(M // 1024, max(1, 512 // tiles_r)) >= 256 and get_tuning("wgrad_rect") != 0:
            name = "gemm_kernel<128,64,false,false>"
    if dt == BF16 and Nout % 128 == 0 and K % 128 == 0 and M % 128 == 0 and (Nout // 128) * (K // 128) * groups >= 64:
        name = "wgrad3_kernel"              # csrc/wgrad.hip: 128x128 tiles, 8 waves
    esz = x.element_size()
    _timed(name, 2.0 * M * Nout * K * groups, groups * (esz * M * Nout + esz * M * K + 4.0 * Nout * K), lambda: call(
        "nsid_linear_bwd_weight", _p(dout), dout.shape[-1], _p(x), x.shape[-1], _p(dw), M, Nout, K, groups,
        _p(in_scale), _p(in_shift), act_in, dt, _stream()), (M, Nout, K, groups))


WGRAD_GROUPED = 1        # 0: the deferred phase issues one nsid_linear_bwd_weight per recorded problem (the gate of round 6: 8.68 ms)


def wgrad_item(it) -> None:
    """one recorded weight-gradient problem through the per-layer launch (the grouped phase's fallback and its test reference)"""
    extra = it[10] if len(it) > 10 else None
    if extra is None:
        linear_bwd_weight(*it[:10])
    else:                                      # ("ds", B, N, C): the packed gradient of a Downsample (unpacked by the caller)
        _, B, N, C = extra[:4]
        downsample3_bwd_weight(it[0], it[1], it[2], B, N, C, it[4])


def linear_bwd_weight_batch(items, max_workgroups: int = 0) -> None:
    """the deferred weight-gradient phase (functional.DeferredWgrads): items = [(dout, x, dw, M, Nout, K, groups, in_scale, in_shift,
    act_in[, extra])] in backward order, both views of a layer sharing the same dw. dw += dout^T f(x) for every item.
    extra = ("ds", B, N, C, weight_grad): a Downsample — x is its input, dw its packed gradient, unpacked into weight_grad afterwards.
    ONE call of nsid_linear_bwd_weight_grouped per storage type: the items that share dw and shape become the two row segments of one
    problem, the problems of all layers one launch per tile class (csrc/gemm.hip wgrad_grouped_kernel).
    max_workgroups > 0: a launch that runs beside the backward chains (at most that many workgroups, each walking several items)."""
    if not items:
        return
    from ._lib import WgradProblem
    import ctypes
    unpack = {}
    for it in items:
        if len(it) > 10 and it[10] is not None:
            unpack[it[2].data_ptr()] = (it[2], it[10][4])
    if not WGRAD_GROUPED:
        for it in items:
            wgrad_item(it)
    else:
        for dtype, code in ((torch.bfloat16, BF16), (torch.float32, F32)):
            probs, open_ = [], {}           # open_: (dw pointer, shape) -> a problem that still has a free second segment
            flops = nbytes = 0.0
            n_items = 0
            for it in items:
                dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in = it[:10]
                if dout.dtype != dtype:
                    continue
                extra = it[10] if len(it) > 10 else None
                _chk(dw, in_scale, in_shift)
                if _act(dout, x) != code:
                    raise RuntimeError("a weight-gradient problem with mixed storage types")
                n_items += 1
                flops += 2.0 * M * Nout * K * groups
                nbytes += groups * float(dout.element_size()) * M * (Nout + K)
                ldx = x.shape[-1]
                key = (dw.data_ptr(), M, Nout, K, groups, dout.shape[-1], ldx, act_in, in_scale is None, extra is None)
                q = open_.pop(key, None)
                if q is None:
                    q = WgradProblem()
                    q.dout[0], q.x[0], q.in_scale[0], q.in_shift[0] = _p(dout), _p(x), _p(in_scale), _p(in_shift)
                    q.dw, q.ldd, q.ldx, q.M, q.Nout, q.K, q.groups, q.act_in = _p(dw), dout.shape[-1], ldx, M, Nout, K, groups, act_in
                    q.ds_out_nodes = 0 if extra is None else extra[2] // 2
                    probs.append(q)
                    open_[key] = q
                    nbytes += 4.0 * groups * Nout * K
                else:
                    q.dout[1], q.x[1], q.in_scale[1], q.in_shift[1] = _p(dout), _p(x), _p(in_scale), _p(in_shift)
            if not probs:
                continue
            arr = (WgradProblem * len(probs))(*probs)
            _timed("wgrad_grouped_kernel", flops, nbytes, lambda: call(
                "nsid_linear_bwd_weight_grouped", ctypes.addressof(arr), len(probs), code, int(max_workgroups), _stream()),
                (n_items, len(probs), 0, 1))
    for dwp, wgrad in unpack.values():
        unpack_ds_wgrad(dwp, wgrad)


def colsum_acc(x, out) -> None:
    _chk(out)
    _tk("colsum_atomic_kernel", float(x.element_size()) * x.numel(), lambda: call(
        "nsid_colsum_acc", _p(x), x.shape[-1], x.shape[0], x.shape[1], _p(out), _act(x), _stream()), (x.shape[0], x.shape[1], 0, 1))


# ------------------------------------------------------------------------------------------------ batch norm
class BNAffine:
    """What a consumer needs to apply a producer's BatchNorm on load, plus what backward needs.
    identity: scale == 1, shift == 0 (an eval-mode BatchNorm folded into its conv): consumers that only normalise skip it,
    consumers that also apply an activation on load use the ones/zeros vectors."""
    __slots__ = ("scale", "shift", "mean", "invstd", "identity")

    def __init__(self, scale, shift, mean=None, invstd=None, identity=False):
        self.scale, self.shift, self.mean, self.invstd, self.identity = scale, shift, mean, invstd, identity


_IDENTITY_AFFINE = {}


def identity_affine(C: int, device) -> BNAffine:
    key = (C, str(device))
    a = _IDENTITY_AFFINE.get(key)
    if a is None:
        a = _IDENTITY_AFFINE[key] = BNAffine(torch.ones(C, device=device), torch.zeros(C, device=device), identity=True)
    return a


# eval-mode BatchNorm folded into the preceding conv: BN(W x + b) = (s W) x + (s b + beta - mean s), s = gamma/sqrt(var+eps).
# Constants of the checkpoint: computed once per layer (torch ops, off the hot path), bf16 shadow included, and kept until a
# version counter of one of the six tensors moves.
_FOLDED = {}      # id(gamma) -> (weakref(gamma), versions, w_fold, b_fold)


def folded_conv_bn(w2d, bias, gamma, beta, running_mean, running_var, eps=BN_EPS, source=None):
    """w2d: (Nout_total, K) weight matrix, or a callable that builds it from `source` (the downsample's packed weight);
    `source` is the tensor whose version counter stands for the weight (default: w2d itself)."""
    import weakref
    src = w2d if source is None else source
    versions = (src.data_ptr(), src._version, None if bias is None else (bias.data_ptr(), bias._version), gamma._version,
                beta._version, running_mean._version, running_var._version, eps, WEIGHT_EPOCH, STATS_EPOCH)
    # under stream capture nothing is INSERTED (the constants would be produced by captured work), but an entry made
    # before the capture is used: the replayed kernels then read the cached constants in place (GraphedFingerprinter)
    cacheable = not torch.cuda.is_current_stream_capturing()
    e = _FOLDED.get(id(gamma))
    if e is not None and e[0]() is gamma and e[1] == versions:
        return e[2], e[3]
    with torch.no_grad():
        if callable(w2d):
            w2d = w2d()
        s = gamma.detach() / torch.sqrt(running_var + eps)
        w_fold = (w2d.detach() * s[:, None]).contiguous()
        b_fold = (beta.detach() - running_mean * s) if bias is None else (bias.detach() * s + beta.detach() - running_mean * s)
        b_fold = b_fold.contiguous()
        if w_fold.numel() % 8 == 0:
            SHADOWS.register(w_fold, f32_to_bf16(w_fold), owner=w_fold, fresh=True)
    if cacheable:
        torch.cuda.current_stream().synchronize()       # once per layer: the constants are then valid on every stream
        if len(_FOLDED) > 4096:
            for k in [k for k, v in _FOLDED.items() if v[0]() is None]:
                del _FOLDED[k]
        _FOLDED[id(gamma)] = (weakref.ref(gamma), versions, w_fold, b_fold)
    return w_fold, b_fold


def bn_finalize(stat, M, gamma, beta, running_mean, running_var, num_batches_tracked,
                momentum=BN_MOMENTUM, eps=BN_EPS) -> BNAffine:
    C = gamma.numel()
    buf = torch.empty((4, C), device=gamma.device, dtype=torch.float32)
    if running_mean is not None:
        bump_state_epoch(weights=False)
    _tk("bn_finalize_kernel", 8.0 * row_tiles(M) * C + 16.0 * C, lambda: call(
        "nsid_bn_finalize", _p(stat), row_tiles(M), C, M, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
        _p(num_batches_tracked), momentum, eps, _p(buf[0]), _p(buf[1]), _p(buf[2]), _p(buf[3]), _stream()), (M, C, 0, 1))
    return BNAffine(buf[0], buf[1], buf[2], buf[3])


# Diagnosis only (results wrong, timing valid): bit 0 skips the forward finalize launches, bit 1 the backward ones — the ceiling of
# what moving the finalize arithmetic into neighbouring kernels can buy (bench.py --flag ops.DIAG_SKIP_FINALIZE=3).
DIAG_SKIP_FINALIZE = 0
_DIAG_CONST = {}


def _diag_const(C, device):
    key = (C, str(device))
    if key not in _DIAG_CONST:
        _DIAG_CONST[key] = (torch.ones(C, device=device), torch.zeros(C, device=device), torch.zeros((2, C), device=device))
    return _DIAG_CONST[key]


def bn_finalize_deferred(stat, M, gamma, beta, eps=BN_EPS):
    """training-mode statistics -> (BNAffine, unbiased variance) WITHOUT updating the running statistics (bn_running_update)"""
    C = gamma.numel()
    if DIAG_SKIP_FINALIZE & 1:
        one, zero, _ = _diag_const(C, gamma.device)
        return BNAffine(one, zero, zero, one), one
    buf = torch.empty((5, C), device=gamma.device, dtype=torch.float32)
    _tk("bn_finalize_kernel", 8.0 * row_tiles(M) * C + 20.0 * C, lambda: call(
        "nsid_bn_finalize_deferred", _p(stat), row_tiles(M), C, M, _p(gamma), _p(beta), eps, _p(buf[0]), _p(buf[1]),
        _p(buf[2]), _p(buf[3]), _p(buf[4]), _stream()), (M, C, 0, 1))
    return BNAffine(buf[0], buf[1], buf[2], buf[3]), buf[4]


def bn_running_update(layers_a, layers_b=None, momentum=BN_MOMENTUM) -> None:
    """Deferred running-statistics updates of many BatchNorm layers in one launch per 16 layers.
    layers_a: [(running_mean, running_var, num_batches_tracked or None, mean, uvar)] of the first view, in layer order;
    layers_b: the same layers for the second view (its mean / uvar are applied after the first view's), or None."""
    import ctypes
    n = len(layers_a)
    if n == 0:
        return
    if layers_b is not None and len(layers_b) != n:
        raise ValueError("both views must have run the same BatchNorm layers")
    bump_state_epoch(weights=False)
    arr = lambda vals: (ctypes.c_void_p * n)(*[None if v is None else v.data_ptr() for v in vals])
    Cs = (ctypes.c_int * n)(*[la[0].numel() for la in layers_a])
    for la in layers_a:
        _chk(la[0], la[1], la[3], la[4])
    if layers_b is not None:
        for la, lb in zip(layers_a, layers_b):
            if la[0].data_ptr() != lb[0].data_ptr():
                raise ValueError("the two views disagree on the layer order")
            _chk(lb[3], lb[4])
    call("nsid_bn_running_update", n, Cs, arr([la[0] for la in layers_a]), arr([la[1] for la in layers_a]),
         arr([la[2] for la in layers_a]), arr([la[3] for la in layers_a]), arr([la[4] for la in layers_a]),
         None if layers_b is None else arr([lb[3] for lb in layers_b]),
         None if layers_b is None else arr([lb[4] for lb in layers_b]), momentum, _stream())


# eval-mode BatchNorm affines are constants of the checkpoint: computed once per layer and kept until one of the four tensors
# changes (torch version counters) — fingerprinting spent 6 % of its GPU time recomputing 64 of them per micro-batch.
_EVAL_AFFINE = {}     # id(gamma) -> (weakref(gamma), versions, eps, BNAffine, stream handle, event)


def bn_eval_affine(gamma, beta, running_mean, running_var, eps=BN_EPS) -> BNAffine:
    import weakref
    versions = (gamma._version, beta._version, running_mean._version, running_var._version,
                beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), WEIGHT_EPOCH, STATS_EPOCH)
    cacheable = not torch.cuda.is_current_stream_capturing()
    e = _EVAL_AFFINE.get(id(gamma))
    if e is not None and e[0]() is gamma and e[1] == versions and e[2] == eps:
        if e[4] != _stream() and cacheable:
            torch.cuda.current_stream().wait_event(e[5])      # produced on another stream (a capture starts after a full
        return e[3]                                           # device synchronisation: nothing to wait for there)
    C = gamma.numel()
    buf = torch.empty((2, C), device=gamma.device, dtype=torch.float32)
    call("nsid_bn_eval_affine", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C, _p(buf[0]),
         _p(buf[1]), _stream())
    aff = BNAffine(buf[0], buf[1])
    if cacheable:
        ev = torch.cuda.Event()
        ev.record()
        if len(_EVAL_AFFINE) > 4096:          # entries of dead models
            for k in [k for k, v in _EVAL_AFFINE.items() if v[0]() is None]:
                del _EVAL_AFFINE[k]
        _EVAL_AFFINE[id(gamma)] = (weakref.ref(gamma), versions, eps, aff, _stream(), ev)
    return aff


def bn_apply(r, aff: BNAffine, act=ACT_NONE, residual=None, out=None) -> torch.Tensor:
    dt = _act(r, residual, out)
    M, C = r.shape
    if out is None:
        out = torch.empty_like(r)
    _tk("bn_apply_kernel", r.element_size() * M * C * (3 if residual is not None else 2), lambda: call(
        "nsid_bn_apply", _p(r), _p(aff.scale), _p(aff.shift), act, _p(residual), _p(out), M, C, dt, _stream()), (M, C, 0, 1))
    return out


def bn_backward(dout, r, aff: BNAffine, act, dgamma, dbeta, inplace=False, partial=None) -> torch.Tensor:
    """Gradient w.r.t. the raw conv output r of y = act(BN(r)), given dL/dy; dgamma/dbeta are accumulated.
    partial: the column sums already produced by the GEMM that wrote dout (linear_bwd_data(bn=...)): skips the reduce."""
    dt = _act(dout, r)
    M, C = r.shape
    tiles = row_tiles(M) if partial is None else partial.shape[1]      # rows of partial sums (one per clip from mr_aggregate_bwd)
    s = _stream()
    coef = torch.empty((2, C), device=r.device, dtype=torch.float32)
    if partial is None:
        partial = torch.empty((2, tiles, C), device=r.device, dtype=torch.float32)
        part_ = partial
        _tk("col_reduce_kernel", 2.0 * r.element_size() * M * C, lambda: call(
            "nsid_bn_bwd_reduce", _p(dout), _p(r), M, C, _p(aff.scale), _p(aff.shift), _p(aff.mean), _p(aff.invstd),
            act, _p(part_), dt, s), (M, C, 0, 1))
    if DIAG_SKIP_FINALIZE & 2:
        coef = _diag_const(C, r.device)[2]
    else:
        _tk("bn_bwd_finalize_kernel", 8.0 * tiles * C + 16.0 * C, lambda: call(
            "nsid_bn_bwd_finalize", _p(partial), tiles, C, M, _p(dgamma), _p(dbeta), _p(coef), s), (M, C, 0, 1))
    dr = dout if inplace else torch.empty_like(dout)
    _tk("bn_bwd_apply_kernel", 3.0 * r.element_size() * M * C, lambda: call(
        "nsid_bn_bwd_apply", _p(dout), _p(r), M, C, _p(aff.scale), _p(aff.shift), _p(aff.mean), _p(aff.invstd),
        act, _p(coef), _p(dr), dt, s), (M, C, 0, 1))
    return dr


# ------------------------------------------------------------------------------------------------ graph ops
def knn_graph(r, B, N, C, k, dilation=1, aff: Optional[BNAffine] = None) -> torch.Tensor:
    """(B*N, C) features (optionally with a pending BatchNorm affine) -> int32 (B, N, k) clip-local neighbour ids"""
    dt = _act(r)
    if aff is not None and aff.identity:
        aff = None
    idx = torch.empty((B, N, k), device=r.device, dtype=torch.int32)
    kd = k * dilation
    sel = 8 < kd <= 64 and N >= get_tuning("knn_sel_min_n")      # csrc/knn.hip nsid_knn_graph
    name_ = "knn2_kernel" if kd <= 8 else ("knn_sel_kernel" if sel else ("knn_rank_kernel" if N <= 128 else "knn_kernel"))
    nraw = _cnt("knn2_raw")
    name = lambda: "knn2_raw_kernel" if _cnt("knn2_raw") > nraw else name_      # (one bf16 MFMA pass: executed flops = algorithmic)
    # read the features once, write int32 ids; 2*N^2*C flop on the fp32 matrix pipe
    _timed(name, 2.0 * B * N * N * C, float(B) * (N * C * r.element_size() + N * k * 4), lambda: call(
        "nsid_knn_graph", _p(r), r.shape[-1], _p(aff.scale) if aff else None, _p(aff.shift) if aff else None,
        B, N, C, k, dilation, _p(idx), dt, _stream()), (B * N, N, C, 1))
    return idx


def mr_aggregate_fwd(r, idx, B, N, C, aff: Optional[BNAffine] = None, want_argmax=True):
    dt = _act(r)
    if aff is not None and aff.identity:
        aff = None
    k = idx.shape[-1]
    u = torch.empty((B * N, 2 * C), device=r.device, dtype=r.dtype)
    amax = torch.empty((B * N, C), device=r.device, dtype=torch.uint8) if want_argmax else None
    # read x once, write the interleaved (x, max-relative) pair, read idx (+ the arg-max bytes for backward)
    nbytes = float(B) * N * (3 * C * r.element_size() + k * 4 + (C if want_argmax else 0))
    _timed("mr_fwd_kernel", 0.0, nbytes, lambda: call(
        "nsid_mr_aggregate_fwd", _p(r), r.shape[-1], _p(aff.scale) if aff else None,
        _p(aff.shift) if aff else None, _p(idx), B, N, C, k, _p(u), _p(amax), dt, _stream()), (B * N, 2 * C, C, 1))
    return u, amax


def mr_bwd_sums_fused(du, N, C, k) -> bool:
    """csrc/mr.hip mrs_launch: does the degree-ranked backward (the form that can also emit BatchNorm column sums) take this launch?"""
    kmin = get_tuning("mr_bwd_sorted_min_k")
    return bool(FUSE_MR_BWD_SUMS and du.dtype == torch.bfloat16 and kmin > 0 and k >= kmin and N * C == 16384 and C & (C - 1) == 0
                and 64 <= C <= 512)


FUSE_MR_BWD_SUMS = True      # the aggregation backward also emits the backward column sums of the BatchNorm in front of it (no reduce launch)


def mr_aggregate_bwd(du, idx, amax, B, N, C, bn=None):
    """dL/dy of the max-relative aggregation. bn = (r, aff, act): y was act(BN(r)) of a conv+BN layer (Grapher fc1); where the
    degree-ranked kernel serves the launch the call returns (dy, partial) with partial[2][B][C] = that BatchNorm's backward column
    sums per clip (bn_backward(partial=...) / bn_backward_linear_bwd_data(partial=...)), else (dy, None)."""
    dt = _act(du)
    k = idx.shape[-1]
    dy = torch.empty((B * N, C), device=du.device, dtype=du.dtype)
    nbytes = float(B) * N * (3 * C * du.element_size() + k * 4 + C)
    if bn is not None and mr_bwd_sums_fused(du, N, C, k):
        r, aff, act = bn
        _chk(aff.scale, aff.shift, aff.mean, aff.invstd)
        partial = torch.empty((2, B, C), device=du.device, dtype=torch.float32)
        rc = [0]

        def launch():
            rc[0] = lib.nsid_mr_aggregate_bwd_bn(_p(du), _p(idx), _p(amax), B, N, C, k, _p(dy), _p(r), r.shape[-1], _p(aff.scale),
                                                 _p(aff.shift), _p(aff.mean), _p(aff.invstd), act, _p(partial), dt, _stream())
        _timed("mr_bwd_sorted_kernel +bn_sums", 0.0, nbytes + float(B) * N * C * r.element_size(), launch, (B * N, C, 2 * C, 1))
        if rc[0] == 0:
            return dy, partial
        # the C side declined (its own LDS / k bounds, which this predicate does not restate): the plain launch + a reduce by the caller
        if PROFILE is not None:
            PROFILE.records.pop()
    c0 = _cnt("mr_bwd_sorted")
    _timed(lambda: "mr_bwd_sorted_kernel" if _cnt("mr_bwd_sorted") > c0 else "mr_bwd_kernel", 0.0, nbytes, lambda: call(
        "nsid_mr_aggregate_bwd", _p(du), _p(idx), _p(amax), B, N, C, k, _p(dy), dt, _stream()),
        (B * N, C, 2 * C, 1))
    return (dy, None) if bn is not None else dy


# ------------------------------------------------------------------------------------------------ downsample
def ds_out_nodes(N: int) -> int:
    return (N - 1) // 2 + 1


def im2col3_fwd(x, B, N, C) -> torch.Tensor:
    dt = _act(x)
    col = torch.empty((B * ds_out_nodes(N), 3 * C), device=x.device, dtype=x.dtype)
    call("nsid_im2col3_fwd", _p(x), B, N, C, _p(col), dt, _stream())
    return col


def im2col3_bwd(dcol, B, N, C) -> torch.Tensor:
    dt = _act(dcol)
    dx = torch.empty((B * N, C), device=dcol.device, dtype=dcol.dtype)
    call("nsid_im2col3_bwd", _p(dcol), B, N, C, _p(dx), dt, _stream())
    return dx


class DsPrep:
    """Step-scoped prepared Downsample weights (csrc/misc.hip ds_prepack_kernel): for every registered conv weight (Cout, Cin, 3, 3) the
    packed forward / backward weights in bf16 and two zeroed packed-gradient buffers (one per view), refreshed by ONE launch at the start
    of a step. A training step spent 36 launches on this (pack, pack_bwd, 2 conversions, zero fill, unpack — per layer and per view);
    7 remain (this one + the unpack of each layer and view, which keeps p.grad complete when backward returns)."""

    def __init__(self):
        self.entries = {}           # weight data_ptr -> dict(w, wp16, wb16, dwp, key)

    MAX_LAYERS = 8                  # csrc/misc.hip DSP_LAYERS

    def purge(self) -> None:
        """drop the layers whose weight is gone or has moved (re-flattened by a newer optimiser): the entries of OTHER live optimisers
        stay (a second FusedClipAdam used to clear them all and silently demoted the first to the per-call launches: ADVICE r4)"""
        for ptr in [k for k, e in self.entries.items() if e["wref"]() is None or e["wref"]().data_ptr() != k]:
            del self.entries[ptr]

    def register(self, w) -> None:
        import weakref
        self.purge()
        if w.data_ptr() in self.entries:
            return
        if len(self.entries) >= self.MAX_LAYERS:
            import warnings
            warnings.warn(f"DsPrep holds {self.MAX_LAYERS} Downsample layers; a further one keeps the per-call pack / convert / zero "
                          "launches (slower, same results)")
            return
        Cout, Cin = w.shape[0], w.shape[1]
        dev = w.device
        self.entries[w.data_ptr()] = dict(wref=weakref.ref(w), Cout=Cout, Cin=Cin, key=None,
                                          wp16=torch.empty((Cout, 3 * Cin), device=dev, dtype=torch.bfloat16),
                                          wb16=torch.empty((2 * Cout, Cin), device=dev, dtype=torch.bfloat16),
                                          dwp=torch.empty((2, Cout, 3 * Cin), device=dev, dtype=torch.float32), slots=0)

    def clear(self) -> None:
        self.entries.clear()

    def _arrays(self, es, *names):
        import ctypes
        n = len(es)
        out = []
        for name in names:
            if name in ("Cout", "Cin"):
                out.append((ctypes.c_int * n)(*[e[name] for e in es]))
            elif name == "w":
                out.append((ctypes.c_void_p * n)(*[e["wref"]().data_ptr() for e in es]))
            else:
                out.append((ctypes.c_void_p * n)(*[e[name].data_ptr() for e in es]))
        return out

    def refresh(self) -> None:
        """pack every registered weight (current values) and zero the shared packed gradients: one launch on the current stream"""
        self.purge()
        es = [e for e in self.entries.values() if e["wref"]().is_cuda] if DS_PREP_ENABLED else []
        if not es:
            return
        w, wp16, wb16, dwp, Cout, Cin = self._arrays(es, "w", "wp16", "wb16", "dwp", "Cout", "Cin")
        _tk("ds_prepack_kernel", sum(4.0 * e["wref"]().numel() / 3 + 2.0 * 5 * e["Cout"] * e["Cin"] + 12.0 * e["Cout"] * e["Cin"] for e in es),
            lambda: call("nsid_ds_prepack", len(es), w, wp16, wb16, dwp, Cout, Cin, _stream()))
        for e in es:
            e["key"], e["slots"] = (e["wref"]()._version, WEIGHT_EPOCH), 0

    def take_slot(self, e):
        """a zeroed packed-gradient buffer of this step (two per layer: one per view), or None when both are taken"""
        if e["slots"] >= e["dwp"].shape[0]:
            return None
        e["slots"] += 1
        return e["dwp"][e["slots"] - 1]

    def get(self, w):
        """the prepared entry of `w` if it was refreshed for the weight's current contents, else None (callers pack per call)"""
        e = self.entries.get(w.data_ptr())
        if not DS_PREP_ENABLED or e is None or e["key"] != (w._version, WEIGHT_EPOCH):
            return None
        owner = e["wref"]()
        # the entry is keyed by address and holds only a weak reference: a NEW weight at the address of a freed one must not inherit
        # its packed copies. `.data` / detach aliases of the registered parameter share its storage and are accepted.
        if owner is None or (owner is not w and (owner.data_ptr() != w.data_ptr() or owner.shape != w.shape or
                                                 owner.untyped_storage().data_ptr() != w.untyped_storage().data_ptr())):
            return None
        return e


DS_PREP = DsPrep()
DS_PREP_ENABLED = 1        # bench.py --flag ops.DS_PREP_ENABLED=0: the per-call pack / convert / zero launches instead (one-box A/B)


def downsample3_fwd(x, B, N, C, wp, bias, Cout, want_stat=False, w16=None):
    """Conv2d 3x3 s2 p1 on a width-1 map as ONE GEMM over a zero-padded strided view of x (no im2col): (B*N, C) ->
    (B*N/2, Cout) raw conv output (+ BatchNorm partial statistics). wp: packed weight (Cout, 3C) fp32 (pack_ds_weight); w16: the same
    already in bf16 (DsPrep; wp may then be None)."""
    _chk(wp, bias)
    dt = _act(x)
    No = N // 2
    M = B * No
    out = torch.empty((M, Cout), device=x.device, dtype=x.dtype)
    stat = torch.empty((2, row_tiles(M), Cout), device=x.device, dtype=torch.float32) if want_stat else None
    wop, wdt = (w16, BF16) if (w16 is not None and dt == BF16) else _weight(wp, dt, 3 * C)      # w16: prepared bf16 packed weight (DsPrep)
    esz = x.element_size()
    _timed("gemm_kernel<128,%d,true,true>" % (64 if Cout <= 64 else 128), 2.0 * M * Cout * 3 * C,
           esz * B * N * C + float(wop.element_size()) * Cout * 3 * C + esz * M * Cout, lambda: call(
        "nsid_downsample3_fwd", _p(x), B, N, C, _p(wop), wdt, _p(bias), _p(out), Cout, _p(stat), dt, _stream()),
        (M, Cout, 3 * C, 1))
    return out, stat


def downsample3_bwd_weight(dout, x, dwp, B, N, C, Cout) -> None:
    """dwp (Cout, 3C) += dout^T col(x), col read as a view of x"""
    _chk(dwp)
    dt = _act(dout, x)
    M = B * (N // 2)
    esz = x.element_size()
    _timed("gemm_kernel<64,64,false,false>", 2.0 * M * Cout * 3 * C, esz * (M * Cout + B * N * C) + 4.0 * Cout * 3 * C,
           lambda: call("nsid_downsample3_bwd_weight", _p(dout), _p(x), _p(dwp), B, N, C, Cout, dt, _stream()),
           (M, Cout, 3 * C, 1))


def downsample3_bwd_data(dout, wp, w_odd, B, N, C, Cout, w16=None) -> torch.Tensor:
    """dx (B*N, C) from dout (B*N/2, Cout): even rows one GEMM, odd rows one GEMM over overlapping rows of dout (no col2im).
    w16 = (packed forward weight, packed odd-row weight) already in bf16 (DsPrep): wp / w_odd may then be None"""
    dt = _act(dout)
    dx = torch.empty((B * N, C), device=dout.device, dtype=dout.dtype)
    if w16 is not None and dt == BF16:
        wop, oop, wdt, odt = w16[0], w16[1], BF16, BF16
    else:
        _chk(wp, w_odd)
        wop, wdt = _weight(wp, dt, 3 * C)
        oop, odt = _weight(w_odd, dt, C)
    if odt != wdt:                       # both from the same arithmetic mode
        wop, wdt, oop, odt = wp, F32, w_odd, F32
    M = B * (N // 2)
    esz = dout.element_size()
    _timed("gemm_kernel<128,%d,true,false>" % (64 if C <= 64 else 128), 2.0 * M * Cout * 3 * C,
           esz * (M * Cout + B * N * C) + float(wop.element_size()) * Cout * 3 * C, lambda: call(
        "nsid_downsample3_bwd_data", _p(dout), _p(wop), _p(oop), wdt, _p(dx), B, N, C, Cout, dt, _stream()),
        (M, Cout, 3 * C, 1))
    return dx


def pack_ds_weight_bwd(w) -> torch.Tensor:
    """(Cout, Cin, 3, 3) -> (2*Cout, Cin): [W_2 ; W_0] of kernel column 1, the right operand of the odd-row backward GEMM"""
    _chk(w)
    Cout, Cin = w.shape[0], w.shape[1]
    wb = torch.empty((2 * Cout, Cin), device=w.device, dtype=torch.float32)
    call("nsid_pack_ds_weight_bwd", _p(w), Cout, Cin, _p(wb), _stream())
    return wb


def pack_ds_weight(w) -> torch.Tensor:
    _chk(w)
    Cout, Cin = w.shape[0], w.shape[1]
    wp = torch.empty((Cout, 3 * Cin), device=w.device, dtype=torch.float32)
    call("nsid_pack_ds_weight", _p(w), Cout, Cin, _p(wp), _stream())
    return wp


def unpack_ds_wgrad(dwp, dw) -> None:
    _chk(dwp, dw)
    call("nsid_unpack_ds_wgrad", _p(dwp), dw.shape[0], dw.shape[1], _p(dw), _stream())


# ------------------------------------------------------------------------------------------------ peak extractor
def peak_patchify_fwd(spec, w, bias, pb, pf, out_dtype=torch.float32):
    _chk(spec, w, bias)
    B, H, W = spec.shape
    F = w.shape[0]
    n = (H // pb) * (W // pf)
    out = torch.empty((B * n, F), device=spec.device, dtype=out_dtype)
    minmax = torch.empty((B, 2), device=spec.device, dtype=torch.float32)
    _tk("patchify_fwd_kernel", 4.0 * B * H * W + float(out.element_size()) * B * n * F, lambda: call(
        "nsid_peak_patchify_fwd", _p(spec), _p(w), _p(bias), B, H, W, pb, pf, F, _p(out), F, _p(minmax), _act(out),
        _stream()), (B * n, F, pb * pf * 3, 1))
    return out, minmax


PATCHIFY_BWD_WS = True      # GraFP's patch: per-clip partial sums + one reduce launch (csrc/misc.hip patchify_bwd2_kernel)


def peak_patchify_bwd(spec, minmax, out, dout, pb, pf, dw, dbias) -> None:
    _chk(spec, minmax, dw, dbias)
    B, H, W = spec.shape
    F = out.shape[-1]
    nbytes = 4.0 * B * H * W + 2.0 * out.element_size() * out.numel()
    if PATCHIFY_BWD_WS and (H, W, pb, pf, F) == (64, 128, 4, 8, 8) and out.is_contiguous() and dout.is_contiguous():
        ws = torch.empty((B, 776), device=spec.device, dtype=torch.float32)
        _tk("patchify_bwd2_kernel", nbytes + 8.0 * ws.numel(), lambda: call(
            "nsid_peak_patchify_bwd_ws", _p(spec), _p(minmax), _p(out), _p(dout), F, B, H, W, pb, pf, F, _p(dw), _p(dbias), _p(ws),
            _act(out, dout), _stream()), (out.shape[0], out.shape[1], pb * pf * 3, 1))
        return
    _tk("patchify_bwd_kernel", nbytes, lambda: call(
        "nsid_peak_patchify_bwd", _p(spec), _p(minmax), _p(out), _p(dout), out.shape[-1], B, H, W, pb, pf,
        out.shape[-1], _p(dw), _p(dbias), _act(out, dout), _stream()), (out.shape[0], out.shape[1], pb * pf * 3, 1))


# ------------------------------------------------------------------------------------------------ head
def node_mean_fwd(x, B, N, C) -> torch.Tensor:
    out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    _tk("node_mean_fwd_kernel", float(x.element_size()) * B * N * C, lambda: call(
        "nsid_node_mean_fwd", _p(x), B, N, C, _p(out), _act(x), _stream()), (B * N, C, 0, 1))
    return out


def node_mean_bwd(dout, B, N, C, dtype=torch.float32) -> torch.Tensor:
    _chk(dout)
    dx = torch.empty((B * N, C), device=dout.device, dtype=dtype)
    _tk("node_mean_bwd_kernel", float(dx.element_size()) * B * N * C, lambda: call(
        "nsid_node_mean_bwd", _p(dout), B, N, C, _p(dx), _act(dx), _stream()), (B * N, C, 0, 1))
    return dx


def elu_bwd(dout, out) -> torch.Tensor:
    _chk(dout, out)
    din = torch.empty_like(dout)
    call("nsid_elu_bwd", _p(dout), _p(out), out.numel(), _p(din), _stream())
    return din


def l2norm_fwd(p, eps):
    _chk(p)
    B, d = p.shape
    z = torch.empty_like(p)
    norm = torch.empty((B,), device=p.device, dtype=torch.float32)
    call("nsid_l2norm_fwd", _p(p), B, d, eps, _p(z), _p(norm), _stream())
    return z, norm


def l2norm_bwd(dz, z, norm, eps) -> torch.Tensor:
    _chk(dz, z, norm)
    B, d = z.shape
    dp = torch.empty_like(z)
    call("nsid_l2norm_bwd", _p(dz), _p(z), _p(norm), B, d, eps, _p(dp), _stream())
    return dp


def ntxent_fwd_bwd(z_i, z_j, tau, p0=0, npairs=None, want_grad=True):
    """z_i, z_j: (Bg, d) global embeddings; this rank owns pairs [p0, p0+npairs).
    Returns (loss contribution of the owned rows / (2*Bg), dz_i, dz_j) — dz_* have npairs rows."""
    _chk(z_i, z_j)
    Bg, d = z_i.shape
    npairs = Bg if npairs is None else npairs
    ws = torch.empty((lib.nsid_ntxent_ws_floats(Bg),), device=z_i.device, dtype=torch.float32)
    loss = torch.empty((1,), device=z_i.device, dtype=torch.float32)
    dzi = torch.empty((npairs, d), device=z_i.device, dtype=torch.float32) if want_grad else None
    dzj = torch.empty((npairs, d), device=z_i.device, dtype=torch.float32) if want_grad else None
    # 2 (2B)^2 d flop per pass (similarities; the backward pass recomputes them), bytes = z in + dz out
    _timed("ntxent_kernels", 2.0 * (2 * Bg) * (2 * npairs) * d * (2 if want_grad else 1), 4.0 * d * (2 * Bg + 2 * npairs), lambda: call(
        "nsid_ntxent_fwd_bwd", _p(z_i), _p(z_j), Bg, d, float(tau), p0, npairs, _p(ws), _p(loss), _p(dzi), _p(dzj),
        _stream()), (2 * npairs, 2 * Bg, d, 1))
    return loss, dzi, dzj


# ------------------------------------------------------------------------------------------------ step plumbing
def fill_zero(t: torch.Tensor) -> torch.Tensor:
    """t[...] = 0 with our own streaming kernel (zero_grad: the captured step holds no ATen kernel)"""
    if not (t.is_cuda and t.is_contiguous()):
        raise RuntimeError("fill_zero needs a contiguous device tensor")
    _tk("fill_zero_kernel", float(t.numel() * t.element_size()), lambda: call(
        "nsid_fill_zero", _p(t), t.numel() * t.element_size(), _stream()), (t.numel(), 1, 0, 1))
    return t


def zeros(shape, device, dtype=torch.float32) -> torch.Tensor:
    return fill_zero(torch.empty(shape, device=device, dtype=dtype))


def scale_f32(x: torch.Tensor, scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x * scale[0] (scale: a device scalar such as autograd's grad_output; None: plain copy)"""
    _chk(x, out)
    if scale is not None:
        scale = scale.reshape(-1)
        _chk(scale)
    if out is None:
        out = torch.empty_like(x)
    call("nsid_scale_f32", _p(x), _p(scale), x.numel(), _p(out), _stream())
    return out


def batched_index_select_fwd(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x (B, C, N) fp32, idx (B, Nq, k) int32 -> (B, C, Nq, k) — the reference's batched_index_select (torch_nn.py:79-98)"""
    _chk(x)
    B, C, N = x.shape
    _, Nq, k = idx.shape
    out = torch.empty((B, C, Nq, k), device=x.device, dtype=torch.float32)
    call("nsid_batched_index_select_fwd", _p(x), _p(idx), B, C, N, Nq, k, _p(out), _stream())
    return out


def batched_index_select_bwd(dout: torch.Tensor, idx: torch.Tensor, N: int) -> torch.Tensor:
    _chk(dout)
    B, C, Nq, k = dout.shape
    dx = zeros((B, C, N), dout.device)
    call("nsid_batched_index_select_bwd", _p(dout), _p(idx), B, C, N, Nq, k, _p(dx), _stream())
    return dx


# ------------------------------------------------------------------------------------------------ optimiser
def sumsq_partial(g_flat) -> torch.Tensor:
    _chk(g_flat)
    n = g_flat.numel()
    partial = torch.empty((lib.nsid_sumsq_blocks(n),), device=g_flat.device, dtype=torch.float32)
    _tk("sumsq_kernel", 4.0 * n, lambda: call("nsid_sumsq_partial", _p(g_flat), n, _p(partial), _stream()), (n, 1, 0, 1))
    return partial


def adam_step(p, g, m, v, hyper, step, partial, gnorm_out) -> None:
    _chk(p, g, m, v, hyper, partial, gnorm_out)
    # reads p, g, m, v and writes p, m, v: 28 bytes per parameter
    _tk("adam_kernel", 28.0 * p.numel(), lambda: call(
        "nsid_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(step), _p(partial), partial.numel(),
        _p(gnorm_out), _stream()), (p.numel(), 1, 0, 1))


# ------------------------------------------------------------------------------------------------ layout
def bcn_to_rows(x, dtype=torch.float32) -> torch.Tensor:
    """(B, C, N[, 1]) fp32 reference layout -> node-major (B*N, C) in the activation storage type"""
    _chk(x)
    B, C, N = x.shape[0], x.shape[1], x.shape[2]
    rows = torch.empty((B * N, C), device=x.device, dtype=dtype)
    call("nsid_bcn_to_rows", _p(x), B, C, N, _p(rows), C, _act(rows), _stream())
    return rows


def rows_to_bcn(rows, B, N) -> torch.Tensor:
    """node-major (B*N, C) of either storage type -> (B, C, N) fp32"""
    C = rows.shape[-1]
    x = torch.empty((B, C, N), device=rows.device, dtype=torch.float32)
    call("nsid_rows_to_bcn", _p(rows), C, B, C, N, _p(x), _act(rows), _stream())
    return x
