"""CPU tests of the flat, mmap-able store format and its converters to/from the reference pickle."""
import os
import pickle

import numpy as np
import pytest

from art_amd.art_index import OpenRetreivalDataStore
from art_amd.flat_store import FlatStore, flat_to_store, store_to_flat, write_flat


def test_roundtrip_sorted_by_id(tmp_path):
    rng = np.random.Generator(np.random.PCG64(1))
    ids = rng.permutation(1000).astype(np.int64) * 3 + 7
    rows = rng.standard_normal((1000, 128)).astype(np.float16)
    path = str(tmp_path / "e.flat")
    write_flat(path, ids, rows)
    fs = FlatStore(path)
    assert len(fs) == 1000 and fs.dim == 128 and fs.dtype_code == 0
    assert np.all(np.diff(fs.ids) > 0)
    order = np.argsort(ids)
    np.testing.assert_array_equal(fs.ids, ids[order].astype(np.int32))
    np.testing.assert_array_equal(np.asarray(fs.rows), rows[order])
    assert os.path.getsize(path) == 64 + 4032 + 1000 * 128 * 2  # header + padded ids + rows
    got = [(i.copy(), r.copy()) for i, r in fs.iter_chunks(300)]
    assert [len(i) for i, _ in got] == [300, 300, 300, 100]
    np.testing.assert_array_equal(np.concatenate([r for _, r in got]), rows[order])


def test_reference_pickle_to_flat_and_back(golden_dir, tmp_path):
    d = os.path.join(golden_dir, "fx_store_pickle")
    st = OpenRetreivalDataStore(os.path.join(d, "merged.pkl"), load_from_path=True, rank=0)
    ref = {int(k): v.copy() for k, v in st.embed_data.items()}
    path = str(tmp_path / "e.flat")
    store_to_flat(st, path)
    fs = FlatStore(path)
    assert list(fs.ids) == sorted(ref.keys())
    for i, row in zip(fs.ids, fs.rows):
        np.testing.assert_array_equal(row, ref[int(i)])
    back = flat_to_store(fs, str(tmp_path / "back.pkl"), rank=0)
    assert type(next(iter(back.embed_data))).__name__ == "int64"
    assert {int(k) for k in back.embed_data} == set(ref)
    for k, v in back.embed_data.items():
        assert v.dtype == np.float16
        np.testing.assert_array_equal(v, ref[int(k)])
    # and it pickles to a file the reference format reader accepts
    back.save_shard()
    back.merge_shards_and_save()
    with open(back.embedding_path, "rb") as f:
        assert list(pickle.load(f).keys()) == ["embed_data"]


def test_errors(tmp_path):
    path = str(tmp_path / "e.flat")
    with pytest.raises(ValueError):
        write_flat(path, np.array([1, 1]), np.zeros((2, 8), np.float16))  # duplicate id
    with pytest.raises(ValueError):
        write_flat(path, np.array([1, 2]), np.zeros((2, 8), np.float32))  # dtype code mismatch
    write_flat(path, np.array([5, 2]), np.ones((2, 8), np.float32), dtype_code=2)
    assert FlatStore(path).dtype_code == 2
    with open(path, "r+b") as f:
        f.truncate(100)
    with pytest.raises(ValueError):
        FlatStore(path)
    with open(path, "wb") as f:
        f.write(b"not a flat store" * 8)
    with pytest.raises(ValueError):
        FlatStore(path)
