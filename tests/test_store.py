"""CPU tests of the evidence store mirror (art_amd.art_index.OpenRetreivalDataStore) against the
fixture produced by the REFERENCE'S OWN class (tests/golden/make_golden.py: art_index.py:16-100 imported by
path): pickle layout, key and value types, float16 cast, merge order, duplicate-id error."""
import json
import os
import pickle

import numpy as np
import pytest

from art_amd.art_index import OpenRetreivalDataStore, detach


@pytest.fixture()
def fx(golden_dir):
    d = os.path.join(golden_dir, "fx_store_pickle")
    with open(os.path.join(d, "meta.json")) as f:
        meta = json.load(f)
    return d, np.load(os.path.join(d, "inputs.npz")), meta


def _make_shards(path, inp):
    stores = []
    for rank in range(3):
        st = OpenRetreivalDataStore(path, load_from_path=False, rank=rank)
        st.add_block_data(inp["ids%d" % rank], inp["emb%d" % rank])
        st.save_shard()
        stores.append(st)
    return stores


def test_reads_reference_pickle(fx):
    d, inp, meta = fx
    st = OpenRetreivalDataStore(os.path.join(d, "merged.pkl"), load_from_path=True, rank=0)
    keys = list(st.embed_data.keys())
    assert [int(k) for k in keys] == inp["merged_keys"].tolist()
    assert type(keys[0]).__name__ == meta["key_type"] == "int64"
    rows = np.asarray(list(st.embed_data.values()))
    assert rows.dtype == np.float16 and list(rows.shape[1:]) == meta["value_shape"]
    np.testing.assert_array_equal(rows, inp["merged_rows"])
    assert st.temp_dir_name == os.path.join(d, "merged_tmp")
    assert set(st.state().keys()) == {"embed_data"}


def test_shard_merge_matches_reference(fx, tmp_path):
    d, inp, meta = fx
    path = str(tmp_path / "evidence.pkl")
    stores = _make_shards(path, inp)
    assert sorted(os.listdir(stores[0].temp_dir_name)) == ["0.pkl", "1.pkl", "2.pkl"]
    listdir_order = os.listdir(stores[0].temp_dir_name)
    stores[0].merge_shards_and_save()
    assert not os.path.exists(stores[0].temp_dir_name)  # the reference removes the shard directory
    # order rule of the reference (art_index.py:77-89): own shard first, the others in listdir order
    expect = inp["ids0"].tolist()
    for fname in listdir_order:
        r = int(fname.split(".")[0])
        if r != 0:
            expect += inp["ids%d" % r].tolist()
    assert [int(k) for k in stores[0].embed_data.keys()] == expect
    # the fixture obeys the same rule with ITS listdir order
    ref_expect = inp["ids0"].tolist()
    for fname in meta["listdir_order"]:
        r = int(fname.split(".")[0])
        if r != 0:
            ref_expect += inp["ids%d" % r].tolist()
    assert inp["merged_keys"].tolist() == ref_expect
    # same content as the reference's merged store, key by key, bit by bit
    ref = dict(zip(inp["merged_keys"].tolist(), inp["merged_rows"]))
    assert len(stores[0].embed_data) == len(ref) == 64
    for k, v in stores[0].embed_data.items():
        assert v.dtype == np.float16
        np.testing.assert_array_equal(v, ref[int(k)])
    # and the file we wrote is a plain {'embed_data': dict} pickle that reloads to the same thing
    with open(path, "rb") as f:
        state = pickle.load(f)
    assert list(state.keys()) == ["embed_data"]
    again = OpenRetreivalDataStore(path, load_from_path=True, rank=0)
    assert [int(k) for k in again.embed_data.keys()] == expect


def test_pickle_bytes_equal_reference_when_order_matches(fx, tmp_path):
    # With the same insertion order our pickle is byte-identical to the reference's file.
    d, inp, meta = fx
    st = OpenRetreivalDataStore(str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    ref = OpenRetreivalDataStore(os.path.join(d, "merged.pkl"), load_from_path=True, rank=0)
    for k, v in ref.embed_data.items():
        st.add_block_data(np.asarray([k], dtype=np.int64), v[None, :])
    os.makedirs(st.temp_dir_name)
    st.save_shard()
    st.merge_shards_and_save()
    assert os.path.getsize(st.embedding_path) == meta["pickle_bytes"]


def test_duplicate_and_overwrite(fx, tmp_path):
    d, inp, meta = fx
    st = OpenRetreivalDataStore(str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    st.add_block_data(inp["ids0"], inp["emb0"])
    with pytest.raises(ValueError) as e:
        st.add_block_data(inp["ids0"][:1], inp["emb0"][:1])
    assert str(e.value) == meta["duplicate_error"]
    st.add_block_data(inp["ids0"][:1], inp["emb0"][1:2], allow_overwrite=True)
    np.testing.assert_array_equal(st.embed_data[inp["ids0"][0]], np.float16(inp["emb0"][1]))
    st.clear()
    assert st.embed_data == {}


def test_merge_asserts(tmp_path, fx):
    d, inp, meta = fx
    path = str(tmp_path / "evidence.pkl")
    stores = _make_shards(path, inp)
    # overlap between shards -> AssertionError (art_index.py:90)
    bad = OpenRetreivalDataStore(path, load_from_path=False, rank=3)
    bad.add_block_data(inp["ids1"][:2], inp["emb1"][:2])
    bad.save_shard()
    with pytest.raises(AssertionError):
        stores[0].merge_shards_and_save()
    # own shard missing -> AssertionError (art_index.py:92)
    path2 = str(tmp_path / "other.pkl")
    a = OpenRetreivalDataStore(path2, load_from_path=False, rank=0)
    b = OpenRetreivalDataStore(path2, load_from_path=False, rank=1)
    b.add_block_data(inp["ids1"], inp["emb1"])
    b.save_shard()
    with pytest.raises(AssertionError):
        a.merge_shards_and_save()


def test_iter_chunks_and_detach(fx):
    import torch
    d, inp, meta = fx
    st = OpenRetreivalDataStore(os.path.join(d, "merged.pkl"), load_from_path=True, rank=0)
    ids, rows = [], []
    for i, r in st.iter_chunks(chunk_rows=10):
        assert r.dtype == np.float16 and i.dtype == np.int64 and len(i) <= 10
        ids.append(i)
        rows.append(r)
    np.testing.assert_array_equal(np.concatenate(ids), inp["merged_keys"])
    np.testing.assert_array_equal(np.concatenate(rows), inp["merged_rows"])
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3).requires_grad_()
    out = detach(t)
    assert isinstance(out, np.ndarray) and out.shape == (2, 3)


def test_constructor_needs_path_outside_megatron(tmp_path):
    with pytest.raises(ValueError):
        OpenRetreivalDataStore(None, load_from_path=False)


def test_constructor_reads_megatron_args_when_present(tmp_path, monkeypatch):
    """Inside an ART process the store falls back to get_args().embedding_path / .rank when it is
    built without a path (art_index.py:20-23) and its progress lines are printed only where
    mpu says so (art_index.py:47,53)."""
    import sys
    import types
    from art_amd.art_index import OpenRetreivalDataStore

    args = types.SimpleNamespace(embedding_path=str(tmp_path / "wiki.pkl"), rank=3)
    mpu = types.SimpleNamespace(is_unitialized=lambda: False, get_data_parallel_rank=lambda: 1)
    fake = types.ModuleType("megatron")
    fake.get_args = lambda: args
    fake.mpu = mpu
    monkeypatch.setitem(sys.modules, "megatron", fake)
    store = OpenRetreivalDataStore(load_from_path=False)
    assert store.embedding_path == args.embedding_path and store.rank == 3
    assert store.temp_dir_name == str(tmp_path / "wiki_tmp")
    store.add_block_data(np.array([5, 6]), np.ones((2, 4), dtype=np.float32))
    store.save_shard()
    assert os.path.exists(os.path.join(store.temp_dir_name, "3.pkl"))
