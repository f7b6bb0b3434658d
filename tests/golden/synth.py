"""Deterministic per-key weight / input synthesis shared by the golden generator and the tests.

The reference's full weights are 74 MB, so nothing but inputs, outputs and checksums is committed:
both sides (the imported reference in `make_golden.py`, and the oracle / HIP modules in `tests/`)
fill a `state_dict` from the same rule, keyed on the parameter NAME, so any module exposing the
reference's key names and shapes gets bit-identical weights.
"""
import zlib

import numpy as np
import torch


def _gen(key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    return g


def synth_tensor(key: str, like: torch.Tensor) -> torch.Tensor:
    """Value for state_dict entry `key` with the shape/dtype of `like`."""
    shape = tuple(like.shape)
    g = _gen(key)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if "relative_pos" in key:
        # dead buffer in the reference (torch_vertex.py:189 passes relative_pos=None); left as is
        return like.detach().clone()
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.1 * torch.randn(shape, generator=g)
    if like.dim() <= 1:
        if key.endswith("weight"):          # BatchNorm gamma
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)   # conv / linear / BN bias
    fan_in = int(np.prod(shape[1:]))
    return torch.randn(shape, generator=g) * fan_in ** -0.5


def synth_state(state: dict, prefix: str = "") -> dict:
    """New state dict with every entry of `state` replaced by its synthesized value.
    `prefix` is prepended to the key before hashing, so a sub-module tested stand-alone
    (e.g. one Grapher) can be given the same numbers it would have inside the full model."""
    return {k: synth_tensor(prefix + k, v) for k, v in state.items()}


def synth_randn(tag: str, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=_gen("input:" + tag))


def synth_clips(batch: int, n_mels: int = 64, n_frames: int = 128):
    """Log-mel-like synthetic clip pair: x_i = randn*20-40, x_j = x_i + 3*randn."""
    gi = torch.Generator().manual_seed(0)
    gj = torch.Generator().manual_seed(1)
    x_i = torch.randn(batch, n_mels, n_frames, generator=gi) * 20.0 - 40.0
    x_j = x_i + 3.0 * torch.randn(batch, n_mels, n_frames, generator=gj)
    return x_i, x_j


GRAFP_CFG = {
    # hot-path keys of config/grafp.yaml (:14-68); the rest of the file is data/augmentation config
    "arch": "grafp", "n_mels": 64, "n_frames": 128, "patch_bins": 4, "patch_frames": 8,
    "n_filters": 8, "bsz_train": 256, "tau": 0.05, "lr": 8.0e-5, "min_lr": 7.0e-7, "T_max": 400,
    "d": 128, "h": 1024, "u": 32, "dim": 2048,
}


def fixed_graph(batch: int, nodes: int, k: int, call: int) -> torch.Tensor:
    """A neighbour table that BOTH sides can write down without a search: idx[b, n, j] = (n + 1 + 7 j + 3 b + 5 call) mod nodes
    (j < k <= 18: the k ids of a row are distinct for every nodes in {32, 64, 128, 256}). Used to teacher-force the training step of the
    deep configuration at the timed batch, where storing the reference's own ids would take 20 MB (tests/golden/make_golden.py::
    gold_deep_b256): training mode is chaotic in the neighbour ids, so both sides must run on the SAME graph — which graph it is does
    not matter for pinning the GEMM / aggregation / BatchNorm / weight-gradient arithmetic at that size."""
    b = torch.arange(batch).view(-1, 1, 1)
    n = torch.arange(nodes).view(1, -1, 1)
    j = torch.arange(k).view(1, 1, -1)
    return (n + 1 + 7 * j + 3 * b + 5 * call) % nodes


def clip_graph_hash(idx) -> np.ndarray:
    """64-bit hash per clip of a whole graph build (B, N, k): every row's neighbour SET (ids sorted, so the search's output order does
    not matter) mixed with the row number, summed over the rows in wrapping uint64. A fixture that stores the reference's ids only on
    its near-tie rows (tests/golden/compact.py::sparse_tape) proves with it that every OTHER row of a rebuilt graph is the reference's."""
    a = np.sort(np.asarray(idx.cpu() if isinstance(idx, torch.Tensor) else idx).astype(np.uint64), axis=-1)
    with np.errstate(over="ignore"):
        code = np.zeros(a.shape[:-1], np.uint64)
        for j in range(a.shape[-1]):
            code = code * np.uint64(257) + a[..., j] + np.uint64(1)
        x = (code + np.arange(a.shape[1], dtype=np.uint64)[None, :] * np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        return x.sum(axis=1, dtype=np.uint64)


def row_set_hash(idx: torch.Tensor) -> torch.Tensor:
    """order-independent 8-bit hash of every row's neighbour SET: sum_j (id_j + 1)^2 mod 251 (uint8). Pins which nodes a search
    selected at a size where the ids themselves are too many to store."""
    v = idx.to(torch.int64) + 1
    return ((v * v).sum(-1) % 251).to(torch.uint8)


def synth_unit_pair(B: int, dim: int = 128):
    """the NT-Xent fixtures' inputs: unit rows z_i, and z_j = unit(z_i + 0.5 noise) (make_golden.py::gold_ntxent)"""
    zi = torch.nn.functional.normalize(synth_randn(f"ntx_i{B}", B, dim), dim=1)
    zj = torch.nn.functional.normalize(zi + 0.5 * synth_randn(f"ntx_j{B}", B, dim), dim=1)
    return zi, zj
