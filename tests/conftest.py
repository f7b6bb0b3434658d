import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from compare import SampledRef  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(dict):
    def t(self, key):
        """the stored tensor; a tests/compare.py::SampledRef for a tensor stored in compact form (maxerr / relerr take either)"""
        v = self[key]
        return v if isinstance(v, SampledRef) else torch.from_numpy(np.asarray(v))


def _by_rule(rule, args):
    """inputs a fixture does not store because a rule reproduces them bit for bit (tests/golden/make_golden.py::save)"""
    from synth import GRAFP_CFG, synth_clips, synth_randn
    if rule == "randn":
        return synth_randn(args[0], *args[1])
    if rule in ("clips_i", "clips_j"):
        return synth_clips(args[0])[0 if rule == "clips_i" else 1]
    if rule in ("unit_i", "unit_j"):
        from synth import synth_unit_pair
        return synth_unit_pair(args[0])[0 if rule == "unit_i" else 1]
    if rule in ("bench_i", "bench_j"):          # bench.py's synth_clips
        gi, gj = torch.Generator().manual_seed(args[1]), torch.Generator().manual_seed(args[1] + 1)
        x_i = torch.randn(args[0], GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gi) * 20.0 - 40.0
        x_j = x_i + 3.0 * torch.randn(args[0], GRAFP_CFG["n_mels"], GRAFP_CFG["n_frames"], generator=gj)
        return x_i if rule == "bench_i" else x_j
    raise KeyError(rule)


def load_golden(name):
    """a fixture of tests/golden as a Golden dict, rule-made inputs regenerated"""
    import json
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        data = {k: z[k] for k in z.files}
    for key in [k[:-2] for k in data if k.endswith("@m")]:          # compact form -> one SampledRef under the tensor's own key
        data[key] = SampledRef(key, data.pop(key + "@s"), data.pop(key + "@c"), data.pop(key + "@m"))
    if data.pop("__gapbits__", None) is not None:                   # margins stored as one bit per row ("< 1e-4"): gap.* as 0 / 1
        for key in [k for k in data if k.startswith("near.")]:
            rows = tuple(data["knnshape." + key[5:]][:-1]) if "knnshape." + key[5:] in data else data["knn." + key[5:]].shape[:-1]
            near = np.unpackbits(data[key])[: int(np.prod(rows))].reshape(rows).astype(bool)
            data["gap." + key[5:]] = np.where(near, np.float32(0.0), np.float32(1.0))
    rules = data.pop("__synth__", None)
    if rules is not None:
        digests = _synth_digests()
        for key, (rule, args) in json.loads(bytes(rules).decode()).items():
            data[key] = _by_rule(rule, args).numpy()
            # the rule must reproduce the bytes the fixture's outputs were computed from: a change of torch's RNG stream would otherwise
            # surface as unexplained parity failures. tests/golden/synth_digests.json is written by make_golden.py.
            want = digests.get(f"{name}/{key}")
            if want is not None:
                import hashlib
                got = hashlib.sha256(np.ascontiguousarray(data[key]).tobytes()).hexdigest()[:16]
                assert got == want, (f"golden {name}: the rule-made input {key!r} ({rule}{args}) no longer reproduces the fixture's bytes "
                                     f"({got} != {want}): torch's RNG stream changed; regenerate with tests/golden/make_golden.py")
    return Golden(data)


_DIGESTS = None


def _synth_digests():
    global _DIGESTS
    if _DIGESTS is None:
        import json
        path = os.path.join(GOLDEN, "synth_digests.json")
        _DIGESTS = json.load(open(path)) if os.path.exists(path) else {}
    return _DIGESTS


@pytest.fixture(scope="session")
def golden():
    return load_golden


def to_rows(x4):
    """reference layout (B,C,N,1) -> node-major (B,N,C)"""
    return x4.squeeze(-1).transpose(1, 2).contiguous()


def from_rows(x3):
    """node-major (B,N,C) -> reference layout (B,C,N,1)"""
    return x3.transpose(1, 2).unsqueeze(-1).contiguous()
