"""The bench line's derived fields (bench.py kernel_table / roofline_entry / measured_step_traffic), on synthetic profiles: no GPU.
VERDICT r3 "what's weak" 11: the table must cover every launch family, the roofline entry must be the dominant kernel over ALL of them,
the kNN matrix fraction is against the fp16 peak with the executed-flop factor, traffic figures come from the committed PMC passes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def prof(**k):
    return {n: {"launches": v[0], "ms": v[1], "flops": v[2], "bytes": v[3]} for n, v in k.items()}


def test_kernel_table_peaks_and_shares():
    p = prof(**{"gemm_kernel<128,128,true,true>": (90, 1.6, 90 * 6.26e9, 90 * 32.2e6), "knn2_kernel": (24, 0.34, 24 * 2.147e9, 24 * 8.6e6),
                "bn_bwd_apply_kernel": (104, 0.8, 0.0, 104 * 50.3e6), "ntxent_kernels": (1, 0.035, 1.3e8, 5.2e5), "adam_kernel": (1, 0.08, 0.0, 5.14e8)})
    t = bench.kernel_table(p, 8.0, "bf16")
    assert set(t) == set(p)
    g = t["gemm_kernel<128,128,true,true>"]
    assert abs(g["mfma_frac"] - (90 * 6.26e9 / 1.6e-3 / 1e12) / bench.BF16_MFMA_PEAK_TFLOPS) < 1e-3
    k = t["knn2_kernel"]                                   # fp16 MFMA peak, three executed terms per product
    alg_tf = 24 * 2.147e9 / 0.34e-3 / 1e12
    assert k["executed_flop_factor"] == 3.0 and abs(k["mfma_frac"] - 3.0 * alg_tf / bench.BF16_MFMA_PEAK_TFLOPS) < 1e-3
    assert k["mfma_frac"] < 0.2                           # (the round-3 table divided by the fp32 matrix peak: 0.70)
    assert t["bn_bwd_apply_kernel"]["mfma_frac"] is None and t["bn_bwd_apply_kernel"]["hbm_frac"] > 0.5
    assert abs(t["ntxent_kernels"]["mfma_frac"] - (1.3e8 / 0.035e-3 / 1e12) / bench.FP32_MFMA_PEAK_TFLOPS) < 1e-3
    assert abs(sum(v["share_of_step"] for v in t.values()) - (1.6 + 0.34 + 0.8 + 0.035 + 0.08) / 8.0) < 2e-3


def test_roofline_entry_is_the_dominant_kernel_over_every_family():
    p = prof(**{"gemm_kernel<128,128,true,true>": (90, 1.0, 90 * 6.26e9, 90 * 32.2e6), "bn_bwd_apply_kernel": (104, 1.4, 0.0, 104 * 50.3e6)})
    r = bench.roofline_entry(p, "bf16", 4.6)
    assert r["kernel"] == "bn_bwd_apply_kernel" and r["bound"] == "hbm"
    assert abs(r["achieved"] - 104 * 50.3e6 / 1.4e-3 / 1e9) < 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    p["gemm_kernel<128,128,true,true>"]["ms"] = 1.6
    r = bench.roofline_entry(p, "bf16", 4.6)
    assert r["kernel"] == "gemm_kernel<128,128,true,true>"
    if r["traffic"] is not None:                          # committed PMC entry: ratio = measured bytes / algorithmic bytes of a launch
        assert abs(r["traffic_ratio"] - r["traffic"] / r["alg_bytes_per_launch"]) < 2e-3 and 0.9 < r["traffic_ratio"] < 2.0
    # strict fp32: the GEMMs are matrix-pipe bound, a streaming kernel still is not
    assert bench.roofline_entry(p, "fp32", 4.6)["bound"] == "mfma"
    p["gemm_kernel<128,128,true,true>"]["ms"] = 1.0
    assert bench.roofline_entry(p, "fp32", 4.6)["bound"] == "hbm"


def test_step_traffic_comes_from_the_committed_pmc_passes():
    st = bench.measured_step_traffic("")
    assert st is not None and 10.0 < st[0] < 60.0 and st[1].startswith("profiles/")
    with open(os.path.join(ROOT, st[1])) as f:
        d = json.load(f)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in d["kernels"].values())
    assert abs(d["step_traffic_GB"] - tot / d["steps_profiled"] / 1e9) < 1e-2       # the figure is the sum of its own table
    mb = bench.measured_step_traffic("_infer", key="microbatch_traffic_GB")
    assert mb is not None and 5.0 < mb[0] < 40.0


def test_rocprof_family_maps_kernel_names_onto_the_launch_families():
    f = bench.rocprof_family
    g = ("void (anonymous namespace)::gemm_kernel<128, 128, true, false, true, true, false, true, true, false, 1, 0, false, 0, 4, %s>"
         "((anonymous namespace)::GemmArgs)")
    assert f(g % "false") == "gemm_kernel<128,128,true,false>" and f(g % "true") == "gemm_kernel<128,128,true,false> +bn_apply_load"
    assert f("void (anonymous namespace)::ws_bwd_kernel<64, 256, 1, false, true, true>((anonymous namespace)::WsBwdArgs)") == "ws_bwd_kernel +bn_apply_load"
    assert f("void (anonymous namespace)::ws_fwd_kernel<128, 128, 1, false, true>((anonymous namespace)::WsArgs)") == "ws_fwd_kernel"
    assert f("_ZN12_GLOBAL__N_120mr_bwd_sorted_kernelILb1EEEvPKDF16bPKiPKhiiiiPDF16bNS_5MrsBnE") == "mr_bwd_sorted_kernel +bn_sums"
    assert f("_ZN12_GLOBAL__N_117mr_fwd_lds_kernelIDF16bEEvPKT_lPKfS5_PKiiiiPS1_Phi") == "mr_fwd_kernel"
    assert f("void (anonymous namespace)::wgrad3_grouped_kernel<true>((anonymous namespace)::WgGroupArgs)") == "wgrad_grouped_kernel"
    assert f("void (anonymous namespace)::wgrad_grouped_kernel<64, 64, false, true, true, 2>((anonymous namespace)::WgGroupArgs)") == "wgrad_grouped_kernel"
    assert f("(anonymous namespace)::ntxent_lse_kernel(float const*, float const*, int, int, float, int, float*, float*, float*)") == "ntxent_kernels"
    assert f("(anonymous namespace)::bn_finalize_kernel(float const*, int, int)") == "bn_finalize_kernel"


def test_the_committed_bench_line_agrees_with_the_committed_rocprof_trace():
    """VERDICT r5 task 5: the kernel table of round 5 summed to 9.1 ms where the rocprofv3 trace of the same tree had 11.4 ms (event
    pairs under-timed short kernels). The round-6 line takes its durations from a live rocprofv3 child run: for the committed pair
    profiles/r06c/bench.json + kernel_stats.csv (two separate runs of one tree on one box) the table's sum over its families must lie
    within 0.90 ... 1.05 of the trace's sum over the same families, family by family within 25 % for everything above 100 us per step."""
    import csv
    with open(os.path.join(ROOT, "profiles", "r06c", "bench.json")) as fh:
        line = json.loads(fh.read().strip().splitlines()[-1])
    assert line["kernel_timing"].startswith("rocprofv3") and line["step_traffic_kind"] == "live" and line["roofline"]["traffic_kind"] == "live"
    table = {k: v["avg_us"] * v["launches"] for k, v in line["kernels"].items()}
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06c", "kernel_stats.csv"))))
    steps = max(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
    trace = {}
    for r in rows:
        fam = bench.rocprof_family(r["Name"])
        trace[fam] = trace.get(fam, 0.0) + float(r["TotalDurationNs"]) / 1e3 / steps
    s_table, s_trace = sum(table.values()), sum(v for k, v in trace.items() if k in table)
    assert 0.90 * s_trace <= s_table <= 1.05 * s_trace, (s_table, s_trace)
    for k, us in table.items():
        if us > 100.0:
            assert abs(us - trace[k]) <= 0.25 * trace[k], (k, us, trace[k])
    assert line["config"]["workload"].startswith("B=256 bf16 k=3 hipGraph 2-stream")      # the distinguishing facts survive a truncation
