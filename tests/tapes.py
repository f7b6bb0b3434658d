"""Neighbour tapes stored sparsely (tests/golden/compact.py::sparse_tape): per graph build the reference's ids on its near-tie rows
only, the near-tie bits, and a 64-bit hash per clip of the whole build. A test rebuilds the reference's graphs layer by layer with
KnnTape(patch=patches_of(g, tag)) — the search's own ids, the stored ones on the near-tie rows — and check_patched proves with the
hashes that every other row is the reference's too (before: "0 hard mismatches" against the stored ids)."""
import numpy as np
import torch

from synth import clip_graph_hash


def n_calls(g, tag):
    return len([k for k in g if k.startswith(f"knnshape.{tag}.")])


def patches_of(g, tag):
    """per graph build: (flat near-tie row numbers, the reference's ids on those rows)"""
    out = []
    for c in range(n_calls(g, tag)):
        b, n, _ = (int(v) for v in g[f"knnshape.{tag}.{c}"])
        rows = np.flatnonzero(np.unpackbits(g[f"near.{tag}.{c}"])[: b * n])
        out.append((torch.from_numpy(rows), torch.from_numpy(g[f"nearids.{tag}.{c}"].astype(np.int64))))
    return out


def check_patched(tape, g, tag):
    """tape: a KnnTape that ran in patch mode. (hard, soft, rows): clip x build pairs whose graph — own search + the stored near-tie
    rows — is NOT the reference's (hash over all rows of the clip), near-tie rows where the own search chose another set, rows"""
    hard = soft = rows = 0
    assert len(tape.recorded) == len(tape.patched) == n_calls(g, tag), (len(tape.recorded), len(tape.patched), n_calls(g, tag))
    for c, (own, used) in enumerate(zip(tape.recorded, tape.patched)):
        bad = clip_graph_hash(used) != g[f"graphhash.{tag}.{c}"]
        if bad.any():
            print(f"kNN build {c} {tuple(used.shape)}: clips {np.flatnonzero(bad)[:8]} differ from the reference outside its near-tie rows")
        hard += int(bad.sum())
        own, used = own.detach().cpu().numpy().astype(np.int64), used.detach().cpu().numpy().astype(np.int64)
        soft += int((np.sort(own, -1) != np.sort(used, -1)).any(-1).sum())
        rows += own.shape[0] * own.shape[1]
    return hard, soft, rows
