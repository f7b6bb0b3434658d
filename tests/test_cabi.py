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


def test_ops_refuse_cpu_tensors(libpath):
    """the product path has no CPU fallback: CPU tensors are rejected, not silently computed elsewhere"""
    import torch
    from neuralsampleid_amd import ops
    with pytest.raises(RuntimeError):
        ops.bn_apply(torch.zeros(4, 4), ops.BNAffine(torch.ones(4), torch.zeros(4)))
