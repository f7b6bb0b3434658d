#!/usr/bin/env python3
"""Contrastive-step throughput of the GraFP (GNN) encoder path on MI355X — BASELINE.json's metric.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = train.py:53-75 on one synthetic batch, inputs resident in HBM: zero_grad -> peak-extract -> GraphEncoder
x2 views -> projector -> NT-Xent (global batch) -> backward -> clip_grad_norm_(1.0) -> Adam.  A "clip" is one
(x_i, x_j) pair.  Every kernel in the step is hand-written HIP from libnsid_hip.so; torch supplies memory, streams,
the autograd tape and torch.distributed.  Per-GPU batch is fixed (weak scaling); world > 1 adds the z all-gather and
a SUM all-reduce of the flat gradient buffer over RCCL.

Prints ONE JSON line on rank 0 (fields: the driver's contract + `roofline` + `cpu_baseline`)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = {  # hot-path keys of config/grafp.yaml
    "arch": "grafp", "n_mels": 64, "n_frames": 128, "patch_bins": 4, "patch_frames": 8, "n_filters": 8,
    "bsz_train": 256, "tau": 0.05, "lr": 8.0e-5, "d": 128, "h": 1024, "u": 32, "dim": 2048,
}
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (= fp32 vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0


def synth_clips(batch, seed, device):
    """x_i = randn*20-40 (log-mel-like dB), x_j = x_i + 3*randn"""
    gi = torch.Generator().manual_seed(seed)
    gj = torch.Generator().manual_seed(seed + 1)
    x_i = torch.randn(batch, CFG["n_mels"], CFG["n_frames"], generator=gi) * 20.0 - 40.0
    x_j = x_i + 3.0 * torch.randn(batch, CFG["n_mels"], CFG["n_frames"], generator=gj)
    return x_i.to(device), x_j.to(device)


def _oracle_setup(k, deep):
    from oracle import ref_torch as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(cores, 16)))      # the GPU box gives one GPU a 16-core share
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    torch.manual_seed(42)
    kw = dict(blocks=[4, 4, 12, 4], use_dilation=True) if deep else {}
    sd = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=k, size="t", **kw)).state_dict()   # weights only
    P = {n: v.clone() for n, v in sd.items() if "relative_pos" not in n}
    plan = R.encoder_plan("t", k, blocks=[4, 4, 12, 4] if deep else None, use_dilation=deep)
    return R, P, plan


def cpu_baseline(k, batch=32, steps=3, deep=False):
    """The oracle (CPU restatement of the reference path, pinned to the reference's goldens) timed on this host."""
    R, P, plan = _oracle_setup(k, deep)
    if deep:
        batch, steps = 16, 2                           # 24 blocks at k = 18: keeps the sample within ~20 s
    opt = R.AdamState({n: P[n] for n in R.trainable_keys(P)}, lr=CFG["lr"])
    x_i, x_j = synth_clips(batch, 0, "cpu")
    R.train_step(P, x_i, x_j, CFG, plan, opt)          # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        R.train_step(P, x_i, x_j, CFG, plan, opt)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch / dt, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full contrastive steps at batch {batch} (fp32, oracle/ref_torch.py"
                      f"{', deep plan' if deep else ''}), {dt:.2f} s/step"}


def cpu_baseline_infer(k, batch=128, reps=3, deep=False):
    """forward-only, eval-mode BatchNorm, one view: the oracle's restatement of generate.py:31-49 on this host"""
    R, P, plan = _oracle_setup(k, deep)
    x, _ = synth_clips(batch, 0, "cpu")
    with torch.no_grad():
        fwd = lambda: R.projector(R.graph_encoder(R.peak_patchify(x, P, "peak_extractor.", CFG), P, "encoder.", plan,
                                                  False, None), P, "projector.")
        fwd()
        t0 = time.perf_counter()
        for _ in range(reps):
            fwd()
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(batch / dt, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} eval-mode forward passes of {batch} clips (fp32, oracle/ref_torch.py), {dt:.2f} s each"}


KNN_EXECUTED_FLOP_FACTOR = 3.0    # csrc/knn.hip: a distance product is three fp16 MFMA terms (a a + a b + b a of the two-part split)


def kernel_table(prof, elapsed_ms, precision):
    """per-kernel summary of an ops.KernelProfile: every launch family of the step that ops.py instruments (GEMMs, kNN, aggregation,
    BatchNorm family, patchify, NT-Xent, optimiser, conversions). `tflops` counts ALGORITHMIC flops; mfma_frac divides the
    EXECUTED matrix flops by the dense peak of the operand type the kernel issues:
      GEMM family: bf16 MFMA (2.5 PF) or exact-fp32 MFMA (157 TF) by --precision, executed = algorithmic;
      kNN: fp16 MFMA (same 2.5 PF dense peak as bf16), executed = 3 x algorithmic (three-term split product) — the fast kNN paths
      issue fp16 MFMAs in BOTH precision modes (round 3 divided by the fp32 matrix peak and printed 0.70 for a kernel at 0.13);
      NT-Xent: exact-fp32 MFMA."""
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if precision == "bf16" else FP32_MFMA_PEAK_TFLOPS
    kernels = {}
    for n, d in prof.items():
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        e = {"launches": d["launches"], "avg_us": round(1e3 * d["ms"] / d["launches"], 2), "tflops": round(tf, 2), "mfma_frac": None,
             "alg_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBPS, 4), "share_of_step": round(d["ms"] / elapsed_ms, 3)}
        if d["flops"] > 0:
            if n.startswith("knn2_raw"):          # one bf16 MFMA pass on the stored features: executed = algorithmic
                e["mfma_frac"] = round(tf / BF16_MFMA_PEAK_TFLOPS, 4)
            elif n.startswith("knn"):
                e["executed_flop_factor"] = KNN_EXECUTED_FLOP_FACTOR
                e["mfma_frac"] = round(tf * KNN_EXECUTED_FLOP_FACTOR / BF16_MFMA_PEAK_TFLOPS, 4)
            elif n.startswith("ntxent"):
                e["mfma_frac"] = round(tf / FP32_MFMA_PEAK_TFLOPS, 4)
            else:
                e["mfma_frac"] = round(tf / mfma_peak, 4)
        kernels[n] = e
    return kernels


def roofline_entry(prof, precision, bracket_us, tag="", live=None):
    """the dominant kernel of the instrumented step — over EVERY launch family ops.py times, not only the GEMMs.
    live: the result of live_profile() (prof then already carries its rocprofv3 durations, apply_live_timing)"""
    dom = max(prof, key=lambda n: prof[n]["ms"])
    d = prof[dom]
    alg = d["bytes"] / d["launches"]
    tr, kind = None, None
    if live is not None and live.get("traffic") and dom in live["traffic"]["per_launch"]:
        # a family's PMC bytes per KERNEL launch x kernels per recorded launch of the family
        per_rec = live["launches"].get(dom, d["launches"]) / d["launches"]
        tr = (int(live["traffic"]["per_launch"][dom] * per_rec), "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this bench invocation")
        kind = "live"
    if tr is None:
        tr = measured_traffic(dom, precision, tag)
        kind = "committed" if tr else None
    timed_live = live is not None and dom in live.get("per_step_us", {})
    common = {"kernel": dom, "traffic": tr[0] if tr else None,
              # fabric-side bytes the PMC passes measured per launch over the algorithmic bytes of that launch: > 1 = re-reads
              "traffic_ratio": round(tr[0] / alg, 3) if (tr and alg > 0) else None,
              "traffic_source": (tr[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)") if tr else None,
              "traffic_kind": kind,        # "live": measured in this run; "committed": looked up in profiles/ by kernel name
              "launches_per_step": d["launches"],
              "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
              "flops_per_launch": round(d["flops"] / d["launches"]),
              "alg_bytes_per_launch": round(d["bytes"] / d["launches"]),
              "event_bracket_us": round(bracket_us, 2),
              "method": ("rocprofv3 --kernel-trace --stats of an eager single-stream step of this configuration, run as a child process of "
                         "this bench invocation after the timed region (HIP events, bracket-corrected, remain only for families rocprofv3 "
                         "does not name)") if timed_live else
                        ("HIP events around every launch in one instrumented eager single-stream step after the timed region, "
                         "enqueued behind a spin kernel so that the launches run back to back (an idle GPU would add the host's "
                         "launch gap to every pair), minus the median duration of an empty event bracket (event_bracket_us)")}
    is_gemm = dom.startswith(("gemm_kernel", "gemm256", "wgrad3"))
    if precision == "fp32" and is_gemm:      # fp32 MFMA runs at 1/16 of the bf16 rate: the GEMMs are matrix-pipe bound
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), **common}
    # bf16 operands: every layer's intensity (<= 195 flop/B) is under the 312 flop/B ridge -> HBM is the roof
    ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), **common}


def infer_bench(args, model, rank, world, dev, dist):
    """config 5: forward-only, eval-mode BN, contiguous shard per rank, no collective on the data path"""
    from neuralsampleid_amd import fingerprint
    lo, hi = fingerprint.shard_bounds(args.clips, rank, world)
    mb = args.micro_batch
    n_mb = (hi - lo + mb - 1) // mb
    out = torch.empty((mb, CFG["d"]), device=dev)
    front, waves = None, None
    if args.from_wave:                                            # 4 waveforms of exactly `mb` segments each
        from neuralsampleid_amd.frontend import LogMelFrontEnd
        fcfg = {"fs": 16000, "n_fft": 1024, "win_len": 1024, "hop_len": 512, "n_mels": CFG["n_mels"],
                "n_frames": CFG["n_frames"], "overlap": 0.875}
        front = LogMelFrontEnd(fcfg, dev)
        frames = (mb - 1) * front.step + CFG["n_frames"]
        g = torch.Generator().manual_seed(99 + rank)
        waves = [(0.1 * torch.randn((frames - 1) * 512 + 8, generator=g)).to(dev) for _ in range(4)]
        pool = torch.cat([front(w) for w in waves])
        assert pool.shape[0] == 4 * mb
    else:
        pool, _ = synth_clips(4 * mb, 77 + rank, dev)             # synthetic clips, reused round-robin
    for i in range(max(1, args.warmup)):
        fingerprint.extract_fingerprints(model, pool[:mb], mb, out)
    torch.cuda.synchronize()
    graphed = None
    if not args.no_graph:            # one hipGraph replay per micro-batch: an eager loop of ~130 launches is host-bound
        try:
            graphed = fingerprint.GraphedFingerprinter(model, mb, streams=args.infer_streams)
            graphed(pool[:mb], out)
            torch.cuda.synchronize()
        except Exception as e:
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphed = None
    extract = (lambda x, o: graphed(x, o)) if graphed is not None else \
        (lambda x, o: fingerprint.extract_fingerprints(model, x, mb, o))
    # the extraction call takes many micro-batches at once (the fingerprinter deals them over its streams): calls of 12
    big = big_out = None
    if front is None and graphed is not None:
        # the whole shard resident in HBM (100 000 clips = 3.3 GB of fp32 log-mel, 51 MB of fingerprints): ONE extraction call
        reps = (hi - lo + pool.shape[0] - 1) // pool.shape[0]
        big = pool.repeat(reps, 1, 1)[:hi - lo]
        big_out = torch.empty((big.shape[0], CFG["d"]), device=dev)
        graphed(big[:4 * mb], big_out[:4 * mb])                   # warm every stream once
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    done = 0
    if big is not None:
        while done < hi - lo:
            n = min(big.shape[0], hi - lo - done)
            graphed(big[:n], big_out[:n])
            done += n
    for i in range(n_mb if big is None else 0):
        n = min(mb, hi - lo - done)
        s = (i % 4) * mb
        if front is not None:                                     # waveform -> log-mel segments inside the timed region
            segs = front(waves[i % 4])
            extract(segs[:n], out[:n])
            done += n
            continue
        extract(pool[s:s + n], out[:n])
        done += n
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)          # host-side group (gloo): max over ranks
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    roofline = kernels = cpu = None
    if rank == 0 and not args.no_roofline:
        from neuralsampleid_amd import ops
        ops.PROFILE = ops.KernelProfile()
        bracket_us = 1e3 * ops.PROFILE.bracket_ms
        ops.KernelProfile.plug(20.0)
        fingerprint.extract_fingerprints(model, pool[:mb], mb, out)
        prof = ops.PROFILE.summary()
        ops.PROFILE = None
        kernels = kernel_table(prof, 1e3 * elapsed / n_mb, args.precision)
        roofline = roofline_entry(prof, args.precision, bracket_us, "_infer")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (oracle, eval forward) ...")
        cpu = cpu_baseline_infer(args.k, deep=args.deep)
    result = None
    if rank == 0:
        # 9.8 MB per view-clip forward at 2 B/element with every conv output materialised once
        fwd_bytes = 9.8e6 * (1.0 if args.precision == "bf16" else 2.0) * args.clips
        dealt = ""
        if graphed is not None:          # --from-wave hands over one waveform (= one micro-batch) per call: one stream in use
            dealt = (f" (one hipGraph replay each, dealt over {graphed.n_streams} HIP streams)" if big is not None
                     else " (one hipGraph replay per extraction call)")
        result = {
            "metric": "audio clips/sec (forward-only fingerprint extraction, grafp encoder)",
            "value": round(args.clips / elapsed, 1), "unit": "clips/s", "n_gpus": world, "steps": n_mb,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / n_mb, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.clips} clips mb={mb} {args.precision} k={args.k}: fingerprint inference, synthetic clips"
                                   f"{' from 16 kHz waveforms (log-mel front end on the GPU)' if args.from_wave else ''}"
                                   f", eval-mode BN, micro-batch {mb}{dealt}, "
                                   f"GraphEncoder('t', k={args.k}{', deep' if args.deep else ''})",
                       "parallelism": f"shard{world}", "tuning": getattr(args, "tuning", None) or None},
            "roofline": roofline, "step_hbm_frac_algorithmic": round(fwd_bytes / elapsed / 1e9 / HBM_PEAK_GBPS, 4),
            "kernels": kernels, "cpu_baseline": cpu}
        mt = measured_step_traffic("_infer", key="microbatch_traffic_GB") if args.precision == "bf16" else None
        if mt is not None:       # fabric-side GB per 2 048-clip micro-batch from the committed PMC passes, against 9.8 MB x clips
            result["microbatch_traffic_GB"] = mt[0]
            # against the FUSED forward plan's algorithmic bytes (x in / out per launch, weights once per micro-batch), not the
            # train-mode figure of SURVEY 8d (every conv output materialised): fusion made that one meaningless here
            fused = fused_plan_bytes(2048, args.k, deep=args.deep)
            result["microbatch_traffic_ratio"] = round(mt[0] * 1e9 / fused, 3)
            result["microbatch_algorithmic_GB"] = round(fused / 1e9, 3)
            result["microbatch_traffic_source"] = mt[1] + " (micro-batch 2 048)"
    if world > 1:
        dist.barrier()
    return result


def fused_plan_bytes(clips, k=3, deep=False, n_params=18.4e6):
    """Algorithmic HBM bytes of one eval-mode bf16 micro-batch under the fused forward plan (functional.py, fold_eval): per block
    fc1 (x in, r1 out), kNN (r1 in, ids out), aggregation + grouped conv (r1 + ids in, r2 out: 2C wide), fc2 + shortcut (r2 + x in,
    x1 out), FFN in one launch (x1 in, x2 out); Downsample (x in, half the nodes at twice the width out); the log-mel clip in, stem,
    node mean; every weight once per micro-batch as bf16. A node-major tensor of one clip is N * C * 2 = 32 KB at every stage."""
    T = 64 * 256 * 2.0                                   # one (N, C) clip tensor in bf16
    blocks = [4, 4, 12, 4] if deep else [2, 2, 6, 2]
    nodes = [256, 128, 64, 32]
    per_clip = 64 * 128 * 4.0 + 256 * 8 * 2.0 * 2 + T     # spectrogram in, patches out and in again, stem out
    for nb, n in zip(blocks, nodes):
        ids = n * k * 4.0
        per_clip += nb * ((T + T) + (T + ids) + (T + ids + 2 * T) + (2 * T + T + T) + (T + T))
    per_clip += 3 * (T + T) + T + 1024 * 4.0 * 2 + 128 * 4.0
    return per_clip * clips + 2.0 * n_params * (2 if deep else 1)


def measured_traffic(kernel, precision, tag=""):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*/hbm_traffic*.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench, gfx950 FETCH_SIZE x2 correction applied), or None.
    Keys there come from tools/hbm_traffic.py: `gemm_kernel<BM, BN, AR, BR, bf16>` for the GEMM template (its first five
    arguments), the plain kernel name for everything else (wgrad3_kernel, knn2_kernel, ...)."""
    import glob
    import re
    norm = lambda n: re.sub(r"\s+", "", n)
    want = norm(kernel)
    if want.startswith("gemm_kernel<"):
        want = want[:-1] + "," + ("true" if precision == "bf16" else "false") + ">"
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"hbm_traffic{tag}.json"))):
        try:
            with open(path) as f:
                ks = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        fam = {}
        try:
            with open(path) as f:
                fam = json.load(f).get("family", {})
        except (OSError, ValueError):
            pass
        if want in fam:                   # launch-weighted over the instantiations of a templated kernel (tools/hbm_traffic.py)
            best = (int(fam[want]["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT))
            continue
        for name, v in ks.items():
            n = norm(name)
            if n == want or (not want.startswith("gemm_kernel<") and n.split("<")[0] == want.split("<")[0]):
                best = (int(v["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT))
    if best is None:
        log(f"no committed PMC traffic entry matches kernel '{kernel}' (roofline.traffic = null)")
    return best


def measured_step_traffic(tag="", key="step_traffic_GB"):
    """fabric-side GB per step (or per micro-batch) from the newest committed PMC passes (tools/hbm_traffic.py writes the key), or None"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"hbm_traffic{tag}.json"))):
        try:
            with open(path) as f:
                v = json.load(f).get(key)
        except (OSError, ValueError):
            continue
        if v:
            best = (float(v), os.path.relpath(path, ROOT))
    return best


def rocprof_family(name):
    """the kernel-table family (ops.py's launch names) a rocprofv3 kernel name belongs to, or None (kernels that belong to no timed family)"""
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n).strip()
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)([A-Za-z0-9_]+)", n)
    targs = None
    if m:                                             # a mangled name: <length><identifier>I<template arguments>E...
        base = m.group(2)[:int(m.group(1))]
        rest = m.group(2)[int(m.group(1)):]
        targs = rest
    else:
        n = re.sub(r"\(.*$", "", n)                   # drop the parameter list
        mm = re.match(r"([A-Za-z0-9_]+)(<(.*)>)?$", n)
        if not mm:
            return None
        base, targs = mm.group(1), mm.group(3)
    args = [a.strip() for a in targs.split(",")] if (targs and not m) else []
    if base == "gemm_kernel" and len(args) >= 4:
        fam = f"gemm_kernel<{args[0]},{args[1]},{args[2]},{args[3]}>"
        return fam + (" +bn_apply_load" if len(args) >= 16 and args[15] == "true" else "")
    if base == "ws_bwd_kernel":
        return "ws_bwd_kernel" + (" +bn_apply_load" if len(args) >= 6 and args[5] == "true" else "")
    if base == "mr_bwd_sorted_kernel":
        bns = (targs or "").startswith("ILb1") if m else (args[:1] == ["true"])
        return "mr_bwd_sorted_kernel" + (" +bn_sums" if bns else "")
    if base in ("wgrad_grouped_kernel", "wgrad3_grouped_kernel"):
        return "wgrad_grouped_kernel"
    if base.startswith("ntxent_"):
        return "ntxent_kernels"
    alias = {"mr_fwd_lds_kernel": "mr_fwd_kernel", "patchify_bwd2_reduce_kernel": "patchify_bwd2_kernel", "knn2_kernel": "knn2_kernel",
             "bn_finalize_deferred_kernel": "bn_finalize_kernel", "bn_bwd_finalize_fused_kernel": "bn_bwd_finalize_kernel"}
    return alias.get(base, base)


def live_profile(argv, timeout_s=120):
    """Run THIS bench configuration as a child process under `rocprofv3 --kernel-trace --stats` (eager, one stream, 3 steps) and, in two
    more child runs, under `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only beside them). Returns
    {"per_step_us": {family: us}, "launches": {family: n per step}, "steps": n, "traffic": {...} or None} or None when rocprofv3 is
    missing or a child fails: the caller then keeps the HIP-event timing and the committed PMC figures and says so.
    The children are fresh processes (this one has initialised the GPU and never execs)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None or os.environ.get("NSID_BENCH_LIVE_PROFILE", "1") == "0":
        return None
    keep = []
    skip_next = False
    for a in argv:
        if skip_next:
            skip_next = False
            continue
        if a in ("--steps", "--warmup", "--gpus"):
            skip_next = True
            continue
        if a in ("--no-roofline", "--no-cpu-baseline", "--no-other", "--no-graph", "--no-overlap"):
            continue
        keep.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + keep + ["--steps", "3", "--warmup", "1", "--no-graph", "--no-overlap",
                                                                  "--no-cpu-baseline", "--no-roofline", "--no-other"]
    env = dict(os.environ, TMPDIR="/tmp", NSID_BENCH_LIVE_PROFILE="0")
    out = {"traffic": None}
    tmp = tempfile.mkdtemp(prefix="nsid_prof_", dir="/tmp")
    try:
        def run(tag, extra):
            d = os.path.join(tmp, tag)
            cmd = [rp, "--kernel-trace"] + extra + ["--output-format", "csv", "-d", d, "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            if r.returncode != 0:
                raise RuntimeError(f"rocprofv3 child ({tag}) exited {r.returncode}: {r.stderr[-300:]}")
            return d
        d = run("stats", ["--stats"])
        stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            raise RuntimeError("no kernel_stats.csv")
        rows = list(csv.DictReader(open(stats[0])))
        steps = max([int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]] + [0])
        if steps <= 0:
            raise RuntimeError("no optimiser launches in the trace")
        per, cnt, unmatched = {}, {}, 0.0
        for r in rows:
            fam = rocprof_family(r["Name"])
            us = float(r["TotalDurationNs"]) / 1e3 / steps
            if fam is None:
                unmatched += us
                continue
            per[fam] = per.get(fam, 0.0) + us
            cnt[fam] = cnt.get(fam, 0.0) + float(r["Calls"]) / steps
        out.update(per_step_us=per, launches=cnt, steps=steps, kernel_sum_us=sum(per.values()) + unmatched)
        try:                                            # HBM traffic: separate PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hbm_traffic
            fe = hbm_traffic.load(run("fetch", ["--pmc", "FETCH_SIZE"]), "FETCH_SIZE")
            wr = hbm_traffic.load(run("write", ["--pmc", "WRITE_SIZE"]), "WRITE_SIZE")
            fam_bytes, fam_n, total = {}, {}, 0.0
            for k in set(fe) | set(wr):
                f_, w_ = fe.get(k, []), wr.get(k, [])
                n = max(len(f_), len(w_))
                b = (2.0 * sum(f_) + sum(w_)) * 1024.0          # gfx950: FETCH_SIZE reports half of a wide coalesced read
                total += b
                fam = rocprof_family(k) or k
                fam_bytes[fam] = fam_bytes.get(fam, 0.0) + b
                fam_n[fam] = fam_n.get(fam, 0) + n
            nstep = max(len(fe.get("adam_kernel", [])), len(wr.get("adam_kernel", [])), 1)
            out["traffic"] = {"step_GB": total / nstep / 1e9, "per_launch": {k: fam_bytes[k] / max(fam_n[k], 1) for k in fam_bytes}}
        except Exception as e:      # the timing pass stands on its own
            log(f"live PMC passes unavailable ({type(e).__name__}: {e}); traffic figures come from the committed profiles")
        return out
    except Exception as e:
        log(f"live rocprofv3 profile unavailable ({type(e).__name__}: {e}); HIP-event timing and committed PMC figures are used")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def apply_live_timing(kernels, roof_prof, live, elapsed_ms):
    """overwrite the event-timed averages of the kernel table with the live rocprofv3 figures (family by family); returns the profile
    dict (name -> launches / ms / flops / bytes) with the rocprofv3 milliseconds, for roofline_entry"""
    for name, e in kernels.items():
        us = live["per_step_us"].get(name)
        if us is None:
            e["timing"] = "hip events minus the empty-bracket time (no rocprofv3 family of this name)"
            continue
        tf = roof_prof[name]["flops"] / (us * 1e-6) / 1e12
        gbs = roof_prof[name]["bytes"] / (us * 1e-6) / 1e9
        scale = tf / e["tflops"] if e["tflops"] else None
        e.update(avg_us=round(us / e["launches"], 2), tflops=round(tf, 2), alg_GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBPS, 4),
                 share_of_step=round(us * 1e-3 / elapsed_ms, 3), timing="rocprofv3 --kernel-trace --stats (live child run, eager single-stream step)")
        if e.get("mfma_frac") is not None and scale is not None:
            e["mfma_frac"] = round(e["mfma_frac"] * scale, 4)
        roof_prof[name] = dict(roof_prof[name], ms=us * 1e-3)
    return roof_prof


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _run_rank_job(n: int, deadline: float, extra_env: dict):
    """one attempt: n FRESH child ranks through torch.distributed.run; returns (exit code, JSON lines, stdout); 124 = deadline overrun"""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.update(extra_env)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True, env=env)
    try:
        out, _ = proc.communicate(timeout=deadline)
    except subprocess.TimeoutExpired:
        log(f"the {n}-rank job did not finish within {deadline:.0f} s: terminating its process group")
        os.killpg(proc.pid, signal.SIGTERM)
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
        return 124, [], ""
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return proc.returncode, lines, out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes through
    torch.distributed.run — this process has made no GPU call (a process that has initialised the GPU must never exec or be
    replaced) — relay rank 0's single JSON line, and exit non-zero if any rank fails or the job overruns its deadline.

    First multi-GPU contact (task 6): the captured step holds RCCL kernels inside a hipGraph, which has never run with more
    than one rank. If that attempt exits non-zero or overruns its share of the deadline, ONE more set of fresh ranks is started with
    NSID_DP_GRAPH=0 (eager collectives, no capture); the line then says config.rccl.mode = "eager-retry" (first attempt: "graph").
    The retry is skipped when the caller already asked for eager collectives, or with NSID_DP_RETRY=0."""
    log(f"no launcher in the environment: spawning {n} ranks through torch.distributed.run ...")
    deadline = float(os.environ.get("NSID_BENCH_TIMEOUT_S", "1500"))
    may_retry = os.environ.get("NSID_DP_GRAPH", "1") != "0" and os.environ.get("NSID_DP_RETRY", "1") != "0"
    t0 = time.time()
    rc, lines, out = _run_rank_job(n, deadline * (0.6 if may_retry else 1.0), {"NSID_DP_ATTEMPT": "graph"})
    if rc == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    sys.stderr.write(out)
    log(f"launcher exit code {rc}, {len(lines)} JSON lines")
    if not may_retry:
        return rc or 1
    left = deadline - (time.time() - t0)
    if left < 30:
        log("no time left for the eager retry")
        return rc or 1
    log(f"retrying ONCE with eager collectives (NSID_DP_GRAPH=0), {left:.0f} s left")
    rc2, lines2, out2 = _run_rank_job(n, left, {"NSID_DP_GRAPH": "0", "NSID_DP_ATTEMPT": "eager-retry"})
    if rc2 == 0 and len(lines2) == 1:
        print(lines2[0], flush=True)
        return 0
    sys.stderr.write(out2)
    log(f"eager retry: launcher exit code {rc2}, {len(lines2)} JSON lines")
    return rc2 or rc or 1


def child_argv(args):
    """the flags that reproduce this configuration's eager step in a profiling child process (live_profile)"""
    a = ["--batch", str(args.batch), "--k", str(args.k), "--precision", args.precision, "--storage", args.storage]
    if args.deep:
        a.append("--deep")
    for kv in args.tune:
        a += ["--tune", kv]
    for kv in args.flag:
        a += ["--flag", kv]
    return a


def run_config(args, ctx, side=False):
    """one bench line (a dict on rank 0, None elsewhere) for the configuration `args` names; side=True: a secondary configuration
    of a default run (no other-precision side number)"""
    import torch.distributed as dist
    from neuralsampleid_amd import functional as F_
    from neuralsampleid_amd import ops, parallel
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.simclr import SimCLR
    rank, world, dev, tuning, rehearsal = ctx["rank"], ctx["world"], ctx["dev"], ctx["tuning"], ctx["rehearsal"]
    args.tuning = tuning
    torch.manual_seed(42)                                   # identical initial weights on every rank
    enc_kw = dict(blocks=[4, 4, 12, 4], use_dilation=True) if args.deep else {}
    if args.deep:
        args.k = 18
    model = SimCLR(CFG, GraphEncoder(CFG, in_channels=CFG["n_filters"], k=args.k, size="t", **enc_kw),
                   overlap_views=not args.no_overlap).to(dev).train()
    if args.mode == "infer":
        return infer_bench(args, model, rank, world, dev, dist)
    opt = FusedClipAdam(model.parameters(), lr=CFG["lr"], max_norm=1.0)
    x_i, x_j = synth_clips(args.batch, 1000 + 2 * rank, dev)   # each rank owns different clips
    loss_buf = torch.zeros((), device=dev)
    one = torch.ones((), device=dev)          # static grad_output of loss.backward(): autograd then launches no fill kernel
    reducer = parallel.GradReducer(opt.params, opt.flat_g, opt.offsets, bucket_bytes=16 << 20).install()
    parallel.ASYNC_LOSS_REDUCE = True        # reducer.finish() joins the communicator's stream before the loss is read

    def step():
        opt.zero_grad()
        reducer.start_step()
        if model._side_stream is not None:
            reducer.streams = [torch.cuda.current_stream(), model._side_stream]
        _, _, z_i, z_j = model(x_i, x_j)
        loss = parallel.dist_ntxent_loss(z_i, z_j, CFG)      # z all-gather; NT-Xent over the global batch
        loss.backward(one)                                   # bucketed all-reduce fires as buckets complete
        reducer.finish()
        opt.step()
        ops.scale_f32(loss.detach().reshape(1), None, loss_buf.reshape(1))      # own copy kernel: no ATen kernel in the step

    def barrier():
        # a rank that died leaves the others inside ncclAllReduce / a graph replay: bounded wait, then abort + exit(1)
        parallel.sync_with_deadline(what="the bench step")
        if world > 1:
            dist.barrier()                         # gloo, created with an explicit timeout (parallel.init_from_env)
        parallel.sync_with_deadline(what="the bench step")

    # ---- warm-up (eager), then capture the whole step in a hipGraph (launch-bound inner loop: ~1.2k kernels/step)
    n_eager = max(1, min(args.warmup, 3))
    for i in range(n_eager):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            log(f"eager warm-up step {i} done, loss {float(loss_buf):.4f}")
    def capture(tag):
        """the whole step as ONE hipGraph (all ranks capture, or none does); None -> run eagerly"""
        g = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # world > 1: RCCL's helper threads may touch the HIP runtime while we capture; only calls made by the
            # capturing thread may invalidate the capture ("thread_local"), not theirs
            mode = os.environ.get("NSID_CAPTURE_MODE", "thread_local" if world > 1 else "global")
            with torch.cuda.graph(g, capture_error_mode=mode):
                step()
            if rank == 0:
                log(f"{tag} step captured in a hipGraph")
        except Exception as e:        # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            g = None
            torch.cuda.synchronize()
        if world > 1:                 # replayed collectives must match on every rank: all capture, or none
            ok = torch.tensor([1 if g is not None else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok) == 0:
                g = None
        return g

    graph = None
    if world > 1 and os.environ.get("NSID_DP_GRAPH", "1") == "0":      # escape hatch: eager collectives, no capture
        args.no_graph = True
    if not args.no_graph:
        graph = capture(args.precision)
    run = graph.replay if graph is not None else step
    for _ in range(max(0, args.warmup - n_eager)):
        run()

    barrier()
    t0 = time.perf_counter()
    die_at = os.environ.get("NSID_TEST_KILL_RANK")          # tests only: "rank:step[:attempt]" — that rank dies (SIGKILL) inside the timed loop
    for it in range(args.steps):
        if die_at:
            f = die_at.split(":")
            if int(f[0]) == rank and int(f[1]) == it and (len(f) < 3 or f[2] == os.environ.get("NSID_DP_ATTEMPT", "")):
                import signal
                os.kill(os.getpid(), signal.SIGKILL)
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        log(f"timed {args.steps} steps: {1e3 * elapsed / args.steps:.2f} ms/step")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)             # host-side group (gloo)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss_buf)
    per_step_calls = None
    if parallel.COMM is not None:          # collectives one step enqueues (all-gather of z, loss, gradient buckets)
        c0 = parallel.COMM.calls
        step()
        parallel.sync_with_deadline(what="the collective-count step")
        per_step_calls = parallel.COMM.calls - c0

    roofline, kernels = None, None
    if not args.no_roofline:
        # per-launch HIP-event timing of the GEMM family in one instrumented eager step (events on the launch stream).
        # Every rank runs the step (it contains collectives); only rank 0 instruments and reports.
        if rank != 0:
            step()
            torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        ops.PROFILE = ops.KernelProfile()
        bracket_us = 1e3 * ops.PROFILE.bracket_ms
        model.overlap_views = False          # one stream: a launch's events then bracket that kernel alone
        ops.KernelProfile.plug()             # the host runs ahead of the GPU: event pairs bracket kernels, not launch gaps
        step()
        model.overlap_views = not args.no_overlap
        prof = ops.PROFILE.summary()
        if os.environ.get("NSID_BENCH_SHAPES"):
            with open(os.environ["NSID_BENCH_SHAPES"], "w") as f:
                json.dump(ops.PROFILE.by_shape(), f, indent=0)
        ops.PROFILE = None
        kernels = kernel_table(prof, 1e3 * elapsed / args.steps, args.precision)
        # the event table under-reports short kernels (an empty bracket costs more than the events add around a kernel: VERDICT r5:
        # sum 9.1 ms on the line against 11.4 ms in the trace): when rocprofv3 is on the box the same eager step runs once more under it
        live = None
        if world == 1 and not side and not args.no_live_profile:
            torch.cuda.synchronize()
            live = live_profile(child_argv(args))
        if live is not None:
            prof = apply_live_timing(kernels, dict(prof), live, 1e3 * elapsed / args.steps)
            log(f"live rocprofv3 profile: {live['steps']} steps, kernel sum {live['kernel_sum_us'] / 1e3:.2f} ms per step")
        for e in kernels.values():
            if e["hbm_frac"] > 6.3 / 8.0:        # above what the chip sustains from HBM: the operands were cache-resident
                e["hbm_frac_note"] = "above 6.3 TB/s sustained: not an HBM figure (cache-resident operands or event under-timing)"
        roofline = roofline_entry(prof, args.precision, bracket_us, "_deep" if args.deep else "", live)
        live_traffic = live["traffic"] if live else None

    other, deviation = None, None

    def eval_embeddings():
        """eval-mode (running statistics) embeddings and NT-Xent value of the bench's own clips in the arithmetic that is currently set"""
        from neuralsampleid_amd.simclr.ntxent import ntxent_loss
        model.eval()
        try:
            with torch.no_grad():
                _, _, zi, zj = model(x_i, x_j)
                lv = float(ntxent_loss(zi, zj, CFG))
        finally:
            model.train()
        return zi.float().clone(), zj.float().clone(), lv

    if world == 1 and not args.no_roofline and not args.no_other and not side:
        # side number: the same step in the other arithmetic (eager; the GPU time dominates the host time)
        alt = "fp32" if args.precision == "bf16" else "bf16"
        z_head = eval_embeddings()
        ops.set_gemm_precision(alt)
        F_.set_activation_dtype(alt)
        z_alt = eval_embeddings()
        cos = torch.nn.functional.cosine_similarity(torch.cat(z_head[:2]), torch.cat(z_alt[:2]), dim=1)
        deviation = {"what": f"{args.precision} against {alt} arithmetic of this implementation: the same weights and running statistics "
                             f"(after the timed steps), the bench's {args.batch} clip pairs, eval mode, free-running kNN",
                     "abs_dloss": round(abs(z_head[2] - z_alt[2]), 6), "min_cos_z": round(float(cos.min()), 6),
                     "mean_cos_z": round(float(cos.mean()), 6),
                     "training_step_vs_reference": "tests/test_b256_gpu.py: bf16 step 0 at B = 256 against the fp32 REFERENCE golden, "
                                                   "neighbour ids forced: |dloss| 0.046, min cos z 0.977, early-layer gradients 0.6-0.7 "
                                                   "relative L2 (bf16 storage under 64 train-mode BatchNorms); the fp32 path: 9.5e-7"}
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        alt_graph = None if args.no_graph else capture(alt)     # same treatment as the headline: an eager loop of ~1 000
        alt_run = alt_graph.replay if alt_graph is not None else step   # launches times the host, not the GPU
        for _ in range(2):
            alt_run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_alt = max(3, min(args.steps, 10))
        for _ in range(n_alt):
            alt_run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / n_alt
        other = {"precision": alt, "ms_per_step": round(1e3 * dt, 3), "value": round(args.batch / dt, 1),
                 "unit": "clips/s", "hipgraph": alt_graph is not None,
                 "note": "same model continued; fp32 = strict-parity arithmetic (exact fp32 MFMA GEMMs, fp32 storage; kNN distance products from a two-part fp16 split, 3e-7 from fp64 like the reference's own fp32 GEMM)"}
        del alt_graph
        ops.set_gemm_precision(args.precision)
        F_.set_activation_dtype(args.storage)
        log(f"{alt}: {1e3 * dt:.2f} ms/step")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (oracle) ...")
        cpu = cpu_baseline(args.k, deep=args.deep)
        log(f"cpu baseline {cpu['value']} clips/s on {cpu['cores']} threads")

    if rank == 0:
        clips = args.batch * world * args.steps
        ms = 1e3 * elapsed / args.steps
        # 57 MB per clip pair and step at 2 B/element (115 MB for the deep configuration 4)
        step_bytes = (115e6 if args.deep else 57e6) * (1.0 if args.storage == "bf16" else 2.0) * args.batch
        out = {
            "metric": "audio clips/sec (contrastive step, grafp encoder)", "value": round(clips / elapsed, 1),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"B={args.batch} {args.storage} k={args.k}{' deep[4,4,12,4]' if args.deep else ''} "
                                   f"{'hipGraph' if graph is not None else 'eager'} {'2-stream' if not args.no_overlap else '1-stream'}: "
                                   f"grafp.yaml GraphEncoder('t') full contrastive step (fwd x2 views + NT-Xent + bwd + clip + Adam), "
                                   f"synthetic (64,128) log-mel clip pairs per GPU, random-init weights",
                       "global_batch": args.batch * world, "k": args.k, "parallelism": f"dp{world}",
                       "collectives": ("REHEARSAL over torch.distributed/gloo, ranks sharing devices — not a measurement"
                                       if rehearsal else
                                       "direct RCCL on a dedicated HIP stream: z all-gather + bucketed SUM all-reduce "
                                       "of gradients overlapped with backward" + (", captured in the hipGraph"
                                                                                   if graph is not None else "")
                                       if parallel._distributed() else "none (single process)"),
                       "gemm_arithmetic": ("bf16 MFMA operands, fp32 accumulate" if args.precision == "bf16"
                                           else "fp32 operands, fp32 accumulate"),
                       "activation_storage": args.storage,
                       "views": "two HIP streams (parallel graph branches)" if not args.no_overlap else "sequential",
                       "hipgraph": graph is not None, "final_loss": round(final_loss, 5),
                       # which attempt of the self-spawning launcher this line comes from (spawn_ranks): "graph" = first attempt,
                       # "eager-retry" = fresh ranks with NSID_DP_GRAPH=0 after the first attempt died; None = a launcher started us
                       "dp_attempt": os.environ.get("NSID_DP_ATTEMPT") if world > 1 else None,
                       "tuning": tuning or None, "lib": os.environ.get("NSID_LIB") or None,
                       "rccl": ({"ncclCommCount": parallel.COMM.count(), "collectives_per_step": per_step_calls,
                                 "gradient_buckets": len(reducer.bounds), "bucket_bytes": 16 << 20,
                                 "captured_on_every_rank": graph is not None,
                                 # "graph": collectives replayed inside the step's hipGraph; "eager": not captured (asked for, or the
                                 # capture failed on some rank); "eager-retry": the spawner's second attempt after the first one died
                                 "mode": ("graph" if graph is not None else
                                          ("eager-retry" if os.environ.get("NSID_DP_ATTEMPT") == "eager-retry" else "eager"))}
                                if parallel.COMM is not None else None)},
            "roofline": roofline,
            "step_hbm_frac_algorithmic": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "kernels": kernels,
            "other_precision": other,
            "cpu_baseline": cpu,
        }
        st = measured_step_traffic("_deep" if args.deep else "") if args.storage == "bf16" else None
        out["kernel_timing"] = ("rocprofv3 (live child run)" if (kernels and any("rocprofv3" in e.get("timing", "") for e in kernels.values()))
                                else "hip events")
        lt = locals().get("live_traffic")
        if lt:
            out["step_traffic_GB"] = round(lt["step_GB"], 3)
            out["step_traffic_ratio"] = round(lt["step_GB"] * 1e9 / step_bytes, 3)
            out["step_traffic_source"] = "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this bench invocation"
            out["step_traffic_kind"] = "live"
            st = None
        if st is not None:
            out["step_traffic_kind"] = "committed"
            # fabric-side bytes of ONE step from the committed PMC passes (profiles/*/hbm_traffic*.json), against the algorithmic bytes
            out["step_traffic_GB"] = st[0]
            out["step_traffic_ratio"] = round(st[0] * 1e9 / step_bytes, 3)
            out["step_traffic_source"] = st[1]
        if other is not None and args.precision == "bf16":
            # the headline is bf16 storage + bf16 MFMA operands; the reference runs fp32 (train.py:128): its arithmetic's throughput
            # and how far the two arithmetics are apart belong next to the number
            out["config"]["reference_arithmetic"] = {"precision": other["precision"], "ms_per_step": other["ms_per_step"],
                                                     "value": other["value"], "unit": "clips/s"}
        if deviation is not None:
            out["config"]["arithmetic_deviation"] = deviation
        return out
    return None


def main():
    try:      # past four hardware queues the step's streams share a pipe and the two view branches stop overlapping (docs/experiments.md)
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) > 4:
            print("[bench] GPU_MAX_HW_QUEUES > 4 in the environment: expect ~2.4x the step time (measured 7.9 -> 19 ms)", file=sys.stderr)
    except ValueError:
        pass
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=CFG["bsz_train"], help="clips (pairs) per GPU")
    ap.add_argument("--k", type=int, default=3, help="kNN neighbours (GraphEncoder default 3; train.py --k default 5)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="bf16",
                    help="GEMM operand arithmetic: bf16 operands / fp32 storage+accumulate (BASELINE config 2, default) "
                         "or fp32 (exact fp32 MFMA GEMMs and fp32 storage: the strict-parity path; its kNN distances use the split-fp16 product too); the other mode is timed as a side number")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default=None,
                    help="activation storage in HBM (default: bf16 with --precision bf16 = BASELINE config 2's "
                         "'bf16 storage / fp32 accumulate'; fp32 with --precision fp32)")
    ap.add_argument("--mode", choices=["train", "infer"], default="train",
                    help="train: contrastive step (BASELINE config 2/3, default); infer: forward-only fingerprint "
                         "extraction in eval mode (config 5), --clips per job sharded over the ranks")
    ap.add_argument("--clips", type=int, default=100000, help="infer mode: total clips of the job")
    ap.add_argument("--from-wave", action="store_true",
                    help="infer mode: start from synthetic 16 kHz waveforms (log-mel front end on the GPU, SURVEY 8f-4)")
    ap.add_argument("--deep", action="store_true",
                    help="BASELINE config 4: blocks [4,4,12,4], k=18, intended dilation schedule (capped by N)")
    ap.add_argument("--no-overlap", action="store_true", help="run the two views on one stream instead of two")
    ap.add_argument("--micro-batch", type=int, default=2048,
                    help="--mode infer: clips per forward call (config 5: >= 1024; measured on MI355X: 3.5 ms per 1 024 clips at "
                         "2 048 and 4 096 against 4.1 ms at 1 024 — the 1 024-clip launches leave partial rounds of tiles)")
    ap.add_argument("--infer-streams", type=int, default=2,
                    help="--mode infer: concurrent hipGraph replays (one HIP stream each, the caller's stream among them, micro-batches "
                         "dealt round-robin); measured 386 k / 413 k / 403 k / 408 k clips/s with 1 / 2 / 3 / 4")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-profile", action="store_true",
                    help="do not re-run the eager step under rocprofv3 for the kernel table / traffic figures (HIP events and the committed "
                         "PMC summaries are used instead; also NSID_BENCH_LIVE_PROFILE=0)")
    ap.add_argument("--no-other", action="store_true",
                    help="skip the side measurements of a default run (other precision, configs 4 and 5): A/B sweeps use this")
    ap.add_argument("--flag", action="append", default=[], metavar="MODULE.NAME=VALUE",
                    help="set a module-level switch of the host side for this run, e.g. ops.FUSE_BN_BWD_APPLY=0 (one-box A/B); "
                         "recorded in config.flags")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="set a tuning key of the kernel library (include/nsid.h nsid_set_tuning) for this run; recorded in "
                         "config.tuning. The library never reads the environment.")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch.distributed as dist
    from neuralsampleid_amd import ops, parallel
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.optim import FusedClipAdam
    from neuralsampleid_amd.simclr.simclr import SimCLR

    # data-path collectives: direct RCCL communicator (neuralsampleid_amd/rccl.py) on its own HIP stream; the
    # torch.distributed group (gloo) only carries the ncclUniqueId, the host barriers and the max-over-ranks of the time
    # NSID_BENCH_REHEARSAL=gloo: rehearse the multi-rank orchestration on a box with fewer GPUs than ranks (RCCL refuses two
    # ranks on one device): torch.distributed/gloo carries the data-path collectives, ranks share devices, no hipGraph
    rehearsal = os.environ.get("NSID_BENCH_REHEARSAL", "") == "gloo"
    rank, local, world = parallel.init_from_env("gloo" if rehearsal else "rccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if rehearsal:
        local %= torch.cuda.device_count()
        args.no_graph = True
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    tuning = {}
    for kv in args.tune:
        key, _, val = kv.partition("=")
        ops.set_tuning(key, int(val))
        tuning[key] = int(val)
    from neuralsampleid_amd import functional as F_
    for kv in args.flag:
        target, _, val = kv.partition("=")
        mod, _, name = target.partition(".")
        from neuralsampleid_amd import fingerprint as FP_
        module = {"ops": ops, "functional": F_, "fingerprint": FP_}[mod]
        if not hasattr(module, name):
            raise SystemExit(f"--flag: {mod} has no switch {name}")
        setattr(module, name, int(val))
        tuning[target] = int(val)
    if args.storage is None:
        args.storage = "bf16" if args.precision == "bf16" else "fp32"
    if args.storage == "bf16" and args.precision != "bf16":
        raise SystemExit("--storage bf16 needs --precision bf16 (bf16 tensors feed the bf16 MFMA path)")
    ops.set_gemm_precision(args.precision)
    F_.set_activation_dtype(args.storage)
    ctx = {"rank": rank, "world": world, "dev": dev, "tuning": tuning, "rehearsal": rehearsal}
    default_run = (args.mode == "train" and not args.deep and world == 1 and not args.no_other and not args.no_roofline
                   and args.batch == CFG["bsz_train"] and args.precision == "bf16")
    out = run_config(args, ctx)
    if default_run and rank == 0:
        # BASELINE configs 4 and 5 at their NAMED sizes, timed in the same (driver-run) invocation: the deep step at batch 256 and
        # forward-only extraction of 100 000 clips; each entry is a complete bench line of its own (roofline, traffic, cpu_baseline)
        import copy
        others = {}
        for name, changes in (("config2_k5_step", {"k": 5, "no_cpu_baseline": True}),            # train.py:46 trains with --k 5
                              ("config4_deep_step", {"deep": True}),
                              ("config5_fingerprint_100k", {"mode": "infer", "clips": 100000})):
            a2 = copy.copy(args)
            for k_, v_ in changes.items():
                setattr(a2, k_, v_)
            log(f"other configuration: {name}")
            try:
                others[name] = run_config(a2, ctx, side=True)
            except Exception as e:      # a side configuration must never take the headline down
                others[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        out["other_configs"] = others
        # the same two numbers as scalars, at the top level AND inside `config` (the driver's record keeps `config`, `roofline` and
        # `cpu_baseline` whole and only the NAMES of other top-level keys: VERDICT r4)
        c4, c5 = others.get("config4_deep_step", {}), others.get("config5_fingerprint_100k", {})
        out["config2_k5_ms_per_step"] = out["config"]["config2_k5_ms_per_step"] = others.get("config2_k5_step", {}).get("ms_per_step")
        out["config4_ms_per_step"] = out["config"]["config4_ms_per_step"] = c4.get("ms_per_step")
        out["config5_clips_per_s"] = out["config"]["config5_clips_per_s"] = c5.get("value")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
    parallel.shutdown()


if __name__ == "__main__":
    main()
