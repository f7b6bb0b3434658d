"""CPU: the C-ABI library builds, loads, and exports every symbol include/nsid.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "nsid.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nsid_\w+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def libpath():
    from neuralsampleid_amd.build import build_lib
    return build_lib(verbose=False)


def test_exports_match_header(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_covers_header(libpath):
    from neuralsampleid_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()
    assert _lib.lib.nsid_version() >= 1
    assert _lib.lib.nsid_row_tiles(129) == 2        # host-side helpers need no GPU
    assert _lib.lib.nsid_ntxent_ws_floats(256) >= 4 * 256


def test_tuning_table_roundtrip(libpath):
    """nsid_set_tuning / nsid_get_tuning / nsid_reset_tuning: the only way launch heuristics change (the library never reads
    the environment); unknown keys are refused"""
    import subprocess
    from neuralsampleid_amd import _lib
    keys = _lib.tuning_keys()
    assert len(keys) == len(set(keys)) >= 20 and {"g256_min", "g256_train", "ffn256", "knn_strips", "bn_bwd_apply_max_wg"} <= set(keys)
    defaults = {k: _lib.get_tuning(k) for k in keys}
    assert defaults["g256_min"] == 512 and defaults["bn_bwd_apply_max_wg"] == 384
    _lib.set_tuning("g256_min", 7)
    assert _lib.get_tuning("g256_min") == 7
    _lib.reset_tuning()
    assert {k: _lib.get_tuning(k) for k in keys} == defaults
    with pytest.raises(KeyError):
        _lib.set_tuning("no_such_key", 1)
    # no getenv left in the kernel library
    syms = subprocess.run(["nm", "-D", "--undefined-only", libpath], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_ops_refuse_cpu_tensors(libpath):
    """the product path has no CPU fallback: CPU tensors are rejected, not silently computed elsewhere"""
    import torch
    from neuralsampleid_amd import ops
    with pytest.raises(RuntimeError):
        ops.bn_apply(torch.zeros(4, 4), ops.BNAffine(torch.ones(4), torch.zeros(4)))


def _gfx950_disassembly(libpath, tmp_path):
    """llvm-objdump of every gfx950 code object bundled in the library's .hip_fatbin section (one offload bundle per source)"""
    import struct
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{llvm}/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", libpath, str(tmp_path / "null")], check=True)
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
                co = str(tmp_path / f"co{bi}.o")
                with open(co, "wb") as f:
                    f.write(blob[o + eo:o + eo + es])
                texts.append(subprocess.run([f"{llvm}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True,
                                            text=True, check=True).stdout)
    return texts


def test_no_packed_fma_with_a_constant_multiplier(libpath, tmp_path):
    """`v_pk_fma_f32 d, x, 2.0, v[s:s+1] op_sel:[0,0,1] ...` (hipcc's contraction of `s - 2 x` over two tiles) intermittently lost
    its addend in lanes 48-63 on MI355X / ROCm 7.2 (docs/experiments.md, round 3: one wrong kNN row in 0.1-1 % of launches; the
    source now doubles with x + x). A packed FMA whose MULTIPLIER is an inline constant or literal must not come back in any
    kernel; a constant ADDEND (`..., 0`: a canonicalising multiply) is everywhere and has never misbehaved."""
    texts = _gfx950_disassembly(libpath, tmp_path)
    assert len(texts) >= 10                                   # one code object per source file
    const = re.compile(r"-?\d+(\.\d+)?|0x[0-9a-f]+")
    seen, bad = 0, []
    for text in texts:
        for line in text.splitlines():
            if "v_pk_fma_f32" not in line:
                continue
            seen += 1
            body = line.split("//")[0].split(None, 1)[1]
            args = [a.strip() for a in re.split(r",(?![^\[]*\])", body)]
            if any(const.fullmatch(a) for a in args[1:3]):
                bad.append(line.strip()[:120])
    assert seen > 1000 and not bad, bad[:5]


def test_no_kernel_of_the_library_touches_scratch_memory(libpath, tmp_path):
    """register spills: a kernel that spills pays a scratch round trip per wave and, worse, hides a register budget that no longer
    holds (round 5 shipped ws_fwd_kernel<64,64,4,true,*> with 18-24 scratch instructions, mr_bwd_sorted_kernel<true> with 2,
    gemm256_fwd_kernel<2> with 5). Every gfx950 kernel of the shipped library must be free of scratch_load / scratch_store."""
    texts = _gfx950_disassembly(libpath, tmp_path)
    bad, name, kernels = {}, None, 0
    for text in texts:
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                kernels += 1
            elif name and re.search(r"\bscratch_(load|store)", line):
                bad[name] = bad.get(name, 0) + 1
    assert kernels > 300 and not bad, bad


def test_workspace_query(libpath):
    """SURVEY 8b's per-op scratch query: 0 for the ops that live in LDS, the partial-sum buffers for the three that do not"""
    lib = ctypes.CDLL(libpath)
    lib.nsid_workspace_bytes.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long]
    lib.nsid_workspace_bytes.restype = ctypes.c_long
    lib.nsid_row_tiles.restype = ctypes.c_int
    lib.nsid_ntxent_ws_floats.restype = ctypes.c_size_t
    for op in (b"knn_graph", b"mr_aggregate", b"linear", b"downsample3", b"peak_patchify"):
        assert lib.nsid_workspace_bytes(op, 32768, 128) == 0
    assert lib.nsid_workspace_bytes(b"bn_stat", 32768, 512) == 2 * lib.nsid_row_tiles(32768) * 512 * 4
    assert lib.nsid_workspace_bytes(b"ntxent", 2048, 0) == lib.nsid_ntxent_ws_floats(2048) * 4
    assert lib.nsid_workspace_bytes(b"sumsq", 18_562_664, 0) > 0
    assert lib.nsid_workspace_bytes(b"no_such_op", 1, 1) == -1 and lib.nsid_workspace_bytes(None, 1, 1) == -1
