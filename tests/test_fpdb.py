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


FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fpdb")


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_db_files_match_bytes_written_by_the_reference(tmp_path, impl):
    """tests/golden/fpdb/* were written and read back by the REFERENCE's own statements (test_fp.py:120-133 and
    eval.py:154-196, compiled from their syntax trees by tests/golden/make_golden.py::gold_fpdb): neuralsampleid_amd/fpdb.py
    and the restatement oracle/ref_fpdb.py must both produce those bytes and read them the same way."""
    import shutil
    from neuralsampleid_amd import fpdb
    fp = np.load(os.path.join(FIX, "input_chunks.npy"))
    sizes = np.load(os.path.join(FIX, "input_chunk_sizes.npy"))
    chunks = np.split(fp, np.cumsum(sizes)[:-1])
    for fname in ("ref_db", "query_db"):
        with open(os.path.join(FIX, f"{fname}_lookup.json")) as f:
            names = json.load(f)
        if impl == "product":
            assert fpdb.write_fp_db(str(tmp_path), fname, fp, names) == (13, 128)
        else:
            ref_fpdb.write_db(str(tmp_path), fname, list(chunks), names)
        for suffix in (".mm", "_shape.npy", "_lookup.json"):
            assert filecmp.cmp(tmp_path / f"{fname}{suffix}", os.path.join(FIX, f"{fname}{suffix}"), shallow=False), \
                (fname, suffix)
    # reading: on a copy of the REFERENCE-written files (the reader zeroes NaNs in the file it maps)
    rd = tmp_path / "read"
    rd.mkdir()
    for suffix in (".mm", "_shape.npy"):
        shutil.copy(os.path.join(FIX, "ref_db" + suffix), rd / ("ref_db" + suffix))
    load = fpdb.load_memmap_data if impl == "product" else ref_fpdb.load_memmap_data
    assert tuple(load(str(rd), "ref_db", shape_only=True)) == (13, 128)
    data, shape = load(str(rd), "ref_db")
    assert np.array_equal(np.asarray(shape), np.load(os.path.join(FIX, "reader_shape.npy")))
    want = np.load(os.path.join(FIX, "reader_data.npy"))
    assert np.isnan(fp).sum() == 1 and not np.isnan(want).any()          # the fixture holds a NaN; the reader zeroed it
    assert np.array_equal(np.asarray(data), want)
    del data
    data3, shape3 = load(str(rd), "ref_db", append_extra_length=3)
    assert tuple(data3.shape) == tuple(np.load(os.path.join(FIX, "reader_extra3_shape.npy"))) == (16, 128)
