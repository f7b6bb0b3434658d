#!/usr/bin/env python3
"""Static instruction mix of one kernel of a gfx950 assembly listing, per basic block (which blocks loop, how many VALU /
MFMA / LDS / VMEM / SALU instructions each holds). The GEMM tiles here have 8-64 main-loop stages, so prologue/epilogue
instruction counts matter as much as the loop body.
Usage: hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S x.hip -o x.s; python tools/asm_profile.py x.s <symbol substring>"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r"^(\.LBB\S+):", t)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
        continue
    if not t or t.startswith((";", ".")):
        continue
    cur[1].append(t.split(";")[0].strip())
blocks.append(cur)
names = [b[0] for b in blocks]
def cls(i):
    op = i.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_accvgpr",)): return "acc"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    return "other"
tot = {}
print(f"{'block':14s} {'n':>5s} {'valu':>5s} {'mfma':>5s} {'acc':>4s} {'lds':>4s} {'vmem':>5s} {'salu':>5s} {'wait':>4s} {'bar':>3s}  branches")
for bi, (name, ins) in enumerate(blocks):
    c = {}
    for i in ins:
        c[cls(i)] = c.get(cls(i), 0) + 1
        tot[cls(i)] = tot.get(cls(i), 0) + 1
    tg = [i.split()[-1] for i in ins if cls(i) == "br"]
    back = [t for t in tg if t in names and names.index(t) <= bi]
    print(f"{name:14s} {len(ins):5d} {c.get('valu',0):5d} {c.get('mfma',0):5d} {c.get('acc',0):4d} {c.get('lds',0):4d} {c.get('vmem',0):5d} "
          f"{c.get('salu',0):5d} {c.get('wait',0):4d} {c.get('bar',0):3d}  {' '.join(tg)}{'   <-- LOOP' if back else ''}")
print("total", tot)
