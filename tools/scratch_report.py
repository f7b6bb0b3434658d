#!/usr/bin/env python3
"""Kernels of the shipped library that touch scratch memory (register spills): tools/scratch_report.py [lib.so]
Prints `kernel: n scratch instructions` for every gfx950 kernel with at least one scratch_load / scratch_store."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(libpath, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", libpath, os.path.join(tmp, "null")], check=True)
    blob = open(fat, "rb").read()
    texts = []
    for bi, m in enumerate(re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)):
        o = m.start()
        p = o + 32
        for _ in range(struct.unpack_from("<Q", blob, o + 24)[0]):
            eo, es, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and es:
                co = os.path.join(tmp, f"co{bi}.o")
                with open(co, "wb") as f:
                    f.write(blob[o + eo:o + eo + es])
                texts.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True,
                                            check=True).stdout)
    return texts


def scratch_by_kernel(texts):
    out = {}
    for text in texts:
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                continue
            if name and re.search(r"\bscratch_(load|store)", line):
                out[name] = out.get(name, 0) + 1
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuralsampleid_amd",
                                                             "libnsid_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        res = scratch_by_kernel(disassemble(lib, tmp))
    demangle = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.splitlines()
    for (k, n), d in zip(res.items(), demangle):
        print(f"{n:5d}  {d[:200]}")
    print(f"{len(res)} kernels with scratch instructions")
