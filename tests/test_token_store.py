"""CPU tests of the token datasets: the reader and writer against files the
REFERENCE'S OWN MMapIndexedDatasetBuilder wrote and its reader read back (tests/golden/fx_tokens,
tests/golden/make_golden_tokens.py), the vectorised host fetch against the structure of
art_model.py:367-381, and the C ABI of libarttok.so (load + symbols, no compute without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from art_amd import token_store as ts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "fx_tokens")
NAMES = ("passages_bert", "title_bert", "passages_t0", "title_t0")


def _as_lists(x):
    if isinstance(x, (list, tuple)):
        return [_as_lists(v) for v in x]
    return x


@pytest.mark.parametrize("name", NAMES)
def test_reader_matches_reference_reader(name):
    exp = np.load(os.path.join(FX, "expected.npz"))
    ds = ts.MMapTokenDataset(os.path.join(FX, name))
    assert len(ds) == len(exp[name + "_sizes"])
    assert ds.dtype == (np.uint16 if name.endswith("bert") else np.int32)
    np.testing.assert_array_equal(ds.sizes, exp[name + "_sizes"])
    got = np.concatenate([np.asarray(ds[i]).astype(np.int64) for i in range(len(ds))])
    np.testing.assert_array_equal(got, exp[name + "_flat"])
    # slices, negative index, get() with offset
    assert [a.tolist() for a in ds[3:6]] == [ds[3].tolist(), ds[4].tolist(), ds[5].tolist()]
    assert ds[-1].tolist() == ds[len(ds) - 1].tolist()
    if len(ds[0]) > 2:
        assert ds.get(0, offset=1, length=2).tolist() == ds[0].tolist()[1:3]
    with pytest.raises(ValueError):
        ds[0:4:2]


@pytest.mark.parametrize("name", NAMES)
def test_writer_is_byte_identical_to_reference_builder(tmp_path, name):
    ds = ts.MMapTokenDataset(os.path.join(FX, name))
    out = str(tmp_path / name)
    ts.write_mmap_dataset(out, [ds[i] for i in range(len(ds))], dtype=ds.dtype)
    for ext in (".idx", ".bin"):
        assert open(out + ext, "rb").read() == open(os.path.join(FX, name + ext), "rb").read()


def test_get_batch_with_repeats_and_empty_items():
    ds = ts.MMapTokenDataset(os.path.join(FX, "title_bert"))
    idx = [5, 0, 5, 47, 0]  # item 5 is empty
    flat, offs = ds.get_batch(idx)
    assert offs[0] == 0 and len(offs) == 6
    for j, i in enumerate(idx):
        assert flat[offs[j]:offs[j + 1]].tolist() == ds[i].tolist()
    with pytest.raises(IndexError):
        ds.get_batch([48])


def test_host_fetch_equals_reference_loop():
    exp = np.load(os.path.join(FX, "expected.npz"))
    want = json.load(open(os.path.join(FX, "topk_data.json")))
    stores = [ts.MMapTokenDataset(os.path.join(FX, n)) for n in NAMES]
    got = ts.fetch_topk_data_host(exp["topkindex"], *stores)
    assert _as_lists(got) == want
    assert isinstance(got[0], tuple) and isinstance(got[0][1][0], tuple) and isinstance(got[0][1][0][0], list)


def test_bad_magic_is_rejected(tmp_path):
    p = str(tmp_path / "x")
    open(p + ".idx", "wb").write(b"TNTIDX\x00\x00" + b"\x00" * 32)
    open(p + ".bin", "wb").write(b"")
    with pytest.raises(ValueError):
        ts.MMapTokenDataset(p)


def test_arttok_header_and_library_agree():
    text = open(os.path.join(ROOT, "include", "arttok.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(arttok_[a-z_]+)\s*\(", text)))
    assert declared == sorted(ts.TOK_SYMBOLS)
    lib = ctypes.CDLL(ts.TOK_LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    # argument errors come back as a code + message, without touching a GPU
    lib.arttok_create.restype = ctypes.c_int32
    lib.arttok_last_error.restype = ctypes.c_char_p
    assert lib.arttok_create(0, None, 0, 2, None, None, 0, None) == -1
    assert b"NULL" in lib.arttok_last_error()
