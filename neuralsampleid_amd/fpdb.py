"""Fingerprint database files — the on-disk format the reference's retrieval/evaluation code reads.

Reference writer: test_fp.py:120-133 (`create_fp_db` / `create_ref_db` / `create_dummy_db`): for a database `fname`
  {fname}.mm            raw float32, C order, shape (n_segments, d)     (np.memmap mode 'w+')
  {fname}_shape.npy     np.save of the tuple (n_segments, d)            (loads as an int64 array of 2)
  {fname}_lookup.json   list of n_segments strings (song name per segment; queries: "<name>_<index>")
Reference reader: eval.py:154-196 (`load_memmap_data`: memmap 'r+', NaN -> 0 in place, optional extra rows).
Node-matrix dumps: test_fp.py:244-246 — one `{song_id}.npy` per song holding (num_segments, C, N) float32.

The fingerprints themselves come from `fingerprint.extract_fingerprints` (HIP kernels, eval-mode BN); `build_fp_db` streams
them straight into the memmap, and with several ranks every rank writes its contiguous row range of ONE shared file
(`shard_bounds`), so no gather is needed: the file is complete when all ranks have flushed."""
import json
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .fingerprint import shard_bounds


def write_fp_db(output_root_dir: str, fname: str, fp, lookup_table: Sequence[str]) -> Tuple[int, int]:
    """fp: (n, d) float32 array-like (numpy or a CPU/GPU torch tensor). Byte-identical to the reference writer."""
    fp = _to_numpy(fp)
    if fp.ndim != 2:
        raise ValueError("fingerprints must be (n_segments, d)")
    if len(lookup_table) != fp.shape[0]:
        raise ValueError(f"lookup table has {len(lookup_table)} entries for {fp.shape[0]} fingerprints")
    os.makedirs(output_root_dir, exist_ok=True)
    arr_shape = (int(fp.shape[0]), int(fp.shape[1]))
    arr = np.memmap(os.path.join(output_root_dir, f"{fname}.mm"), dtype="float32", mode="w+", shape=arr_shape)
    arr[:] = fp[:]
    arr.flush()
    del arr
    np.save(os.path.join(output_root_dir, f"{fname}_shape.npy"), arr_shape)
    with open(os.path.join(output_root_dir, f"{fname}_lookup.json"), "w") as f:
        json.dump(list(lookup_table), f)
    return arr_shape


def load_memmap_data(source_dir: str, fname: str, append_extra_length: Optional[int] = None, shape_only: bool = False,
                     display: bool = False):
    """Mirror of eval.py:154-196: returns (memmap 'r+', shape array); NaNs are zeroed in place like the reference does."""
    data_shape = np.load(os.path.join(source_dir, fname + "_shape.npy"))
    if shape_only:
        return data_shape
    if append_extra_length:
        data_shape[0] += append_extra_length
    data = np.memmap(os.path.join(source_dir, fname + ".mm"), dtype="float32", mode="r+",
                     shape=(int(data_shape[0]), int(data_shape[1])))
    data[np.isnan(data)] = 0.0
    if display:
        print(f"Load {data_shape[0]:,} items from {os.path.join(source_dir, fname + '.mm')}.")
    return data, data_shape


def load_lookup(source_dir: str, fname: str) -> List[str]:
    with open(os.path.join(source_dir, f"{fname}_lookup.json")) as f:
        return json.load(f)


def write_node_matrices(save_dir: str, matrices: Dict[str, np.ndarray]) -> None:
    """test_fp.py:244-246: one `{song_id}.npy` per song with the pre-projection node matrices (num_segments, C, N)."""
    os.makedirs(save_dir, exist_ok=True)
    for song_id, nm in matrices.items():
        np.save(os.path.join(save_dir, f"{song_id}.npy"), _to_numpy(nm))


def build_fp_db(model, songs: Iterable[Tuple[str, "object"]], output_root_dir: str, fname: str = "ref_db",
                query_style: bool = False, batch: int = 1024, rank: int = 0, world: int = 1,
                barrier=None) -> Tuple[int, int]:
    """songs: iterable of (name, specs) with specs (S, n_mels, n_frames) fp32 on the GPU (the log-mel segments of one
    audio file). Fingerprints are extracted on the HIP path and written in the reference's format.

    world > 1: every rank is handed the SAME `songs` sequence, extracts only its contiguous share of the segment rows and
    writes them into the shared memmap; rank 0 writes the shape and lookup files (`barrier`: torch.distributed's by
    default). Call a barrier before reading."""
    import torch
    from .fingerprint import extract_fingerprints
    songs = list(songs)
    counts = [int(s.shape[0]) for _, s in songs]
    n = sum(counts)
    d = model.projector[-1].out_features
    lookup: List[str] = []
    for idx, ((nm, _), c) in enumerate(zip(songs, counts)):
        lookup.extend([f"{nm}_{idx}" if query_style else nm] * c)          # test_fp.py:110-115
    os.makedirs(output_root_dir, exist_ok=True)
    path = os.path.join(output_root_dir, f"{fname}.mm")
    if rank == 0:
        arr = np.memmap(path, dtype="float32", mode="w+", shape=(n, d))     # creates / sizes the file
        del arr
        np.save(os.path.join(output_root_dir, f"{fname}_shape.npy"), (n, d))
        with open(os.path.join(output_root_dir, f"{fname}_lookup.json"), "w") as f:
            json.dump(lookup, f)
    if world > 1:                                                           # the file exists before anyone maps it
        if barrier is None:
            import torch.distributed as dist
            barrier = dist.barrier
        barrier()
    lo, hi = shard_bounds(n, rank, world)
    arr = np.memmap(path, dtype="float32", mode="r+", shape=(n, d))
    row = 0
    for (_, specs), c in zip(songs, counts):
        a, b = max(lo, row), min(hi, row + c)
        if a < b:
            z = extract_fingerprints(model, specs[a - row:b - row], batch)
            arr[a:b] = z.detach().cpu().numpy()
        row += c
    arr.flush()
    del arr
    return n, d


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)
