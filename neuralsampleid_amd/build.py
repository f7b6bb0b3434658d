"""Builds libnsid_hip.so (the C-ABI kernel library, include/nsid.h) in-tree for gfx950 with hipcc.

    python -m neuralsampleid_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libnsid_hip.so")
OBJ_DIR = os.path.join(PKG, "csrc", "_obj")
SOURCES = ["tuning.hip", "gemm.hip", "gemm256.hip", "wsgemm.hip", "ffn_fused.hip", "ffn256_fused.hip", "mrconv_fused.hip", "wgrad.hip", "bn.hip", "knn.hip", "mr.hip", "ntxent.hip", "misc.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-I", os.path.join(ROOT, "include")]


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def _deps():
    hdrs = [os.path.join(CSRC, "nsid_common.h"), os.path.join(ROOT, "include", "nsid.h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _check_flags() -> None:
    """the product library is built from the sources and FLAGS alone: a -DNSID_* switch (diagnosis builds change results) smuggled in
    through $HIPCC is refused; tools/build_variant.sh is the place for those"""
    if "-DNSID_" in HIPCC or any(f.startswith("-DNSID_") for f in FLAGS):
        raise RuntimeError("refusing to build the product library with a -DNSID_* switch: use tools/build_variant.sh for diagnosis builds")


def _source_hash() -> str:
    import hashlib
    _check_flags()
    # every flag that shapes the code (the include path is the tree itself) and the compiler command
    h = hashlib.sha256((HIPCC + " " + " ".join(f for f in FLAGS if f != os.path.join(ROOT, "include"))).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "nsid_common.h"),
                                                         os.path.join(ROOT, "include", "nsid.h")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_lib(force: bool = False, verbose: bool = True) -> str:
    # the .so travels to the GPU box without its objects and with fresh mtimes: decide by content, not by time
    stamp = LIB + ".srchash"
    digest = _source_hash()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    force = True
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = _deps()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _newer(s, o) or os.path.getmtime(o) < hdr_time:
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KB)")
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
