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
    assert len(keys) == len(set(keys)) >= 20 and {"g256_min", "g256_train", "w8_min", "knn_strips", "bn_bwd_apply_max_wg"} <= set(keys)
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
