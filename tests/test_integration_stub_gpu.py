"""The reference-side ctypes stub printed in INTEGRATION.md (Level 2) is executed verbatim against the reference's golden
neighbour lists: documentation that cannot run is a boundary bug."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def stub_source():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    src = [b for b in blocks if "nsid_knn_graph" in b]
    assert len(src) == 1
    return src[0]


def test_integration_md_knn_stub_runs_against_the_goldens(golden):
    cwd = os.getcwd()
    os.chdir(ROOT)                        # the stub opens "neuralsampleid_amd/libnsid_hip.so" relative to the repo root
    try:
        ns = {}
        exec(compile(stub_source(), "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    g = golden("knn_c64n256")
    x = g.t("x").to("cuda")
    for k, d in ((3, 1), (4, 2), (18, 3)):
        ei = ns["DenseDilatedKnnGraph"](k, d)(x)
        assert ei.shape == (2, 2, 256, k) and ei.dtype == torch.int64
        assert (ei[1].cpu().numpy() == g[f"center_k{k}_d{d}"]).all()
        a = np.sort(ei[0].cpu().numpy(), -1)
        b = np.sort(g[f"idx_k{k}_d{d}"], -1)
        gap = g[f"mingap_k{k}_d{d}"] if d > 1 else g[f"setgap_k{k}_d{d}"]
        bad = (a != b).any(-1)
        assert int((bad & (gap >= 1e-4)).sum()) == 0 and bad.sum() <= 4
    with pytest.raises(RuntimeError):     # the header's limits are enforced, not silently mis-computed
        ns["DenseDilatedKnnGraph"](3, 1)(x[:, :, :250])


def test_header_argument_count_matches_the_stub():
    with open(os.path.join(ROOT, "include", "nsid.h")) as f:
        hdr = f.read()
    proto = re.search(r"int nsid_knn_graph\((.*?)\);", hdr, flags=re.S).group(1)
    n_header = len([a for a in proto.split(",") if a.strip()])
    argt = re.search(r"argtypes = (.*?)\nNSID_F32", stub_source(), flags=re.S).group(1).replace("\\\n", " ")
    ctypes = __import__("ctypes")
    n_stub = len(eval(argt, {"ctypes": ctypes}))
    assert n_header == n_stub == 12
