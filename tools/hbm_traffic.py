#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<round>/hbm_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-overlap
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- (same)
    python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02/hbm_traffic.json

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of a wide coalesced read -> bytes = (2*FETCH + WRITE)*1024.
GEMM instantiations are keyed by their first five template arguments <BM, BN, A_RMAJOR, B_RMAJOR, bf16> (launch-weighted
over the storage / affine / full-tile variants), which is the key bench.py looks up."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc[simplify(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def simplify(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"(gemm_kernel)<([^>]*)>", n)
    if m:
        args = [a.strip() for a in m.group(2).split(",")][:5]
        return "gemm_kernel<" + ", ".join(args) + ">"
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)([A-Za-z0-9_]+)", n)
    if m:
        return m.group(2)[:int(m.group(1))]
    return n


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1 "
                   "--no-graph --no-overlap`; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE "
                   "reports half of a wide coalesced read, MI355X_MICROARCH.md). Infinity-Cache hits are included in both "
                   "counters (they are the L2's fabric-side requests).",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        out["kernels"][k] = {"launches_profiled": max(len(f), len(w)), "FETCH_SIZE_KB_per_launch": round(fk, 1),
                             "WRITE_SIZE_KB_per_launch": round(wk, 1), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    # launch-weighted family entries for templated kernels other than gemm_kernel (ws_fwd_kernel, ws_bwd_kernel, knn2_raw_kernel, ...):
    # bench.py looks a kernel up by its family name
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, v in out["kernels"].items():
        if "<" in k and not k.startswith("gemm_kernel<"):
            a = fam[k.split("<")[0]]
            a[0] += v["FETCH_SIZE_KB_per_launch"] * v["launches_profiled"]
            a[1] += v["WRITE_SIZE_KB_per_launch"] * v["launches_profiled"]
            a[2] += v["launches_profiled"]
    tot_before = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in out["kernels"].values())
    for k, (f_, w_, n_) in fam.items():
        if k not in out["kernels"] and n_ > 0:
            out["family"] = out.get("family", {})
            out["family"][k] = {"launches_profiled": n_, "FETCH_SIZE_KB_per_launch": round(f_ / n_, 1),
                                "WRITE_SIZE_KB_per_launch": round(w_ / n_, 1), "hbm_bytes_per_launch": int((2 * f_ + w_) / n_ * 1024)}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in out["kernels"].values())
    # steps in the profiled run = launches of the optimiser kernel (training) — bench.py reports the per-step total as step_traffic_GB
    steps = out["kernels"].get("adam_kernel", {}).get("launches_profiled", 0)
    if steps:
        out["steps_profiled"] = steps
        out["step_traffic_GB"] = round(tot / steps / 1e9, 3)
        with open(sys.argv[3], "w") as fh:
            json.dump(out, fh, indent=1)
    mbs = 0
    if not steps:      # forward-only extraction: one patchify launch per micro-batch
        mbs = next((v["launches_profiled"] for k, v in out["kernels"].items() if k.startswith("patchify_fwd")), 0)
        if mbs:
            out["microbatches_profiled"] = mbs
            out["microbatch_traffic_GB"] = round(tot / mbs / 1e9, 3)
            with open(sys.argv[3], "w") as fh:
                json.dump(out, fh, indent=1)
    print(f"{len(out['kernels'])} kernels, {tot / 1e9:.2f} GB over the profiled launches" + (f", {tot / steps / 1e9:.2f} GB per step" if steps else "")
          + (f", {tot / mbs / 1e9:.2f} GB per micro-batch" if mbs else ""))


if __name__ == "__main__":
    main()
