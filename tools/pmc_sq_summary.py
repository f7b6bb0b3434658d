#!/usr/bin/env python3
"""Per-kernel matrix-core and LDS figures from ONE rocprofv3 PMC pass of the bench (its own run: --kernel-trace --pmc
SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE):

    mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   # busy matrix-pipe cycles / (kernel cycles x SIMDs):
                 GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), 256 CUs x 4 SIMDs
    mfma_wall  = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024)    # the same against the kernel's wall time at the MAXIMUM
                 clock: a lower bound that does not depend on GRBM_GUI_ACTIVE, which over-counts on dispatches shorter than ~0.3 ms
                 (the guide's caveat; GRBM / 8 / duration read 3-14 "GHz" on these 5-30 us kernels, so that column is gone)
    lds_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                  # share of LDS-array cycles that are conflict replays

Usage: pmc_sq_summary.py counter_collection.csv[.gz] kernel_trace.csv > profiles/<name>/pmc_sq_summary.txt"""
import collections, csv, gzip, re, sys


def simplify(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)([A-Za-z0-9_]+)", n)
    if m:
        return m.group(2)[:int(m.group(1))]
    return re.sub(r"\s+", "", n)


op = gzip.open if sys.argv[1].endswith(".gz") else open
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with op(sys.argv[1], "rt") as f:
    for r in csv.DictReader(f):
        k = simplify(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
dur = collections.defaultdict(float)
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        dur[simplify(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or cnt[k] == 0:
        continue
    cyc = gui / 8.0
    rows.append((dur[k], k, cnt[k], dur[k] / cnt[k] / 1e3, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024),
                 c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                 c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (max(dur[k], 1.0) * 2.4 * 1024)))
rows.sort(reverse=True)
print(f"{'kernel':100s} {'launches':>8s} {'avg us':>8s} {'mfma_busy':>9s} {'mfma_wall':>9s} {'lds_confl':>9s}")
for d, k, n, us, mf, lc, mw in rows[:40]:
    print(f"{k[:100]:100s} {n:8d} {us:8.1f} {mf:9.3f} {mw:9.3f} {lc:9.3f}")
