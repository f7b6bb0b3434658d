"""CPU: the fingerprint-DB files written by neuralsampleid_amd/fpdb.py are byte-identical to what the reference's writer
produces (oracle/ref_fpdb.py restates test_fp.py:120-133 / eval.py:179-196), and round-trip through the reader."""
import filecmp
import json
import os

import numpy as np
import pytest

from oracle import ref_fpdb


def _fps(n, d, seed=0):
    rng = np.random.default_rng(seed)
    fp = rng.standard_normal((n, d)).astype(np.float32)
    return fp / np.linalg.norm(fp, axis=1, keepdims=True)


@pytest.mark.parametrize("n,d,query", [(37, 128, False), (5, 64, True), (1, 128, False)])
def test_db_files_match_reference_writer(tmp_path, n, d, query):
    from neuralsampleid_amd import fpdb
    fp = _fps(n, d)
    names = [f"song{i // 3}_{i // 3}" if query else f"song{i // 3}" for i in range(n)]
    ours, ref = tmp_path / "ours", tmp_path / "ref"
    ours.mkdir(); ref.mkdir()
    assert fpdb.write_fp_db(str(ours), "ref_db", fp, names) == (n, d)
    ref_fpdb.write_db(str(ref), "ref_db", [fp[:n // 2], fp[n // 2:]] if n > 1 else [fp], names)
    for suffix in (".mm", "_shape.npy", "_lookup.json"):
        assert filecmp.cmp(ours / f"ref_db{suffix}", ref / f"ref_db{suffix}", shallow=False), suffix
    data, shape = ref_fpdb.load_memmap_data(str(ours), "ref_db")          # the reference's reader reads our files
    assert tuple(shape) == (n, d) and np.array_equal(np.asarray(data), fp)
    data2, shape2 = fpdb.load_memmap_data(str(ref), "ref_db")             # and ours reads the reference's
    assert tuple(shape2) == (n, d) and np.array_equal(np.asarray(data2), fp)
    assert fpdb.load_lookup(str(ours), "ref_db") == names


def test_reader_semantics(tmp_path):
    """eval.py:186-193: append_extra_length widens the mapping (the file grows, new rows zero); NaNs are zeroed in place"""
    from neuralsampleid_amd import fpdb
    fp = _fps(6, 16)
    fp[2, 3] = np.nan
    fpdb.write_fp_db(str(tmp_path), "q", fp, ["a"] * 6)
    assert tuple(fpdb.load_memmap_data(str(tmp_path), "q", shape_only=True)) == (6, 16)
    data, shape = fpdb.load_memmap_data(str(tmp_path), "q", append_extra_length=4)
    assert tuple(shape) == (10, 16) and data.shape == (10, 16)
    assert data[2, 3] == 0.0 and not np.isnan(data).any() and np.all(np.asarray(data[6:]) == 0)
    ref, _ = ref_fpdb.load_memmap_data(str(tmp_path), "q")
    assert ref[2, 3] == 0.0                                              # the zeroing was written back ('r+')
    with pytest.raises(ValueError):
        fpdb.write_fp_db(str(tmp_path), "bad", fp, ["a"] * 5)


def test_node_matrix_dump(tmp_path):
    from neuralsampleid_amd import fpdb
    mats = {"s1": np.arange(2 * 512 * 32, dtype=np.float32).reshape(2, 512, 32), "s2": np.zeros((1, 512, 32), np.float32)}
    fpdb.write_node_matrices(str(tmp_path / "nm"), mats)
    for k, v in mats.items():
        got = np.load(tmp_path / "nm" / f"{k}.npy")
        assert got.dtype == np.float32 and np.array_equal(got, v)
