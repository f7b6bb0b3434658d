import os
import sys

import numpy as np
import pytest
import torch

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
        return torch.from_numpy(np.asarray(self[key]))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            return Golden({k: z[k] for k in z.files})
    return load


def to_rows(x4):
    """reference layout (B,C,N,1) -> node-major (B,N,C)"""
    return x4.squeeze(-1).transpose(1, 2).contiguous()


def from_rows(x3):
    """node-major (B,N,C) -> reference layout (B,C,N,1)"""
    return x3.transpose(1, 2).unsqueeze(-1).contiguous()
