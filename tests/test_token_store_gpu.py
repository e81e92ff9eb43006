"""GPU tests of the HBM-resident token store (libarttok.so through its C ABI): the padded gather
against the reader, and the rebuilt get_topk structure against what the reference's loop produced
over its own reader (tests/golden/fx_tokens/topk_data.json)."""
import json
import os

import numpy as np
import pytest
import torch

from art_amd import token_store as ts

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "fx_tokens")
NAMES = ("passages_bert", "title_bert", "passages_t0", "title_t0")


def _as_lists(x):
    if isinstance(x, (list, tuple)):
        return [_as_lists(v) for v in x]
    return x


@pytest.mark.parametrize("name", NAMES)
def test_gather_matches_reader(name):
    ds = ts.MMapTokenDataset(os.path.join(FX, name))
    store = ts.DeviceTokenStore(ds, device=0)
    assert store.info()["n_items"] == len(ds) and store.info()["n_tokens"] == int(ds.sizes.sum())
    rng = np.random.Generator(np.random.PCG64(5))
    ids = rng.integers(1, len(ds) + 1, size=(7, 9)).astype(np.int32)
    ids[0, 0] = 6       # the empty title
    ids[1, 1] = 0       # below the id range
    ids[2, 2] = len(ds) + 1  # above it
    tok, lens = store.gather(torch.from_numpy(ids).cuda(), pad=-7)
    tok, lens = tok.cpu().numpy(), lens.cpu().numpy()
    assert tok.shape == (7, 9, int(ds.sizes.max())) and lens.shape == (7, 9)
    for r in range(7):
        for c in range(9):
            i = ids[r, c] - 1
            if i < 0 or i >= len(ds):
                assert lens[r, c] == -1 and (tok[r, c] == -7).all()
                continue
            want = ds[int(i)].astype(np.int64)
            assert lens[r, c] == len(want)
            np.testing.assert_array_equal(tok[r, c, :len(want)], want)
            assert (tok[r, c, len(want):] == -7).all()
    # truncation to max_len
    tok5, lens5 = store.gather(torch.from_numpy(ids).cuda(), max_len=5, pad=0)
    assert tok5.shape[-1] == 5 and int(lens5.max()) <= 5
    np.testing.assert_array_equal(tok5.cpu().numpy()[3, 3, :int(lens5[3, 3])],
                                  ds[int(ids[3, 3] - 1)][:5].astype(np.int64))
    store.close()


def test_fetch_topk_data_equals_reference_loop():
    exp = np.load(os.path.join(FX, "expected.npz"))
    want = json.load(open(os.path.join(FX, "topk_data.json")))
    stores = [ts.DeviceTokenStore(ts.MMapTokenDataset(os.path.join(FX, n)), device=0) for n in NAMES]
    got = ts.fetch_topk_data(torch.from_numpy(exp["topkindex"]).cuda(), *stores)
    assert _as_lists(got) == want
    bad = exp["topkindex"].copy()
    bad[0, 0] = 1000
    with pytest.raises(IndexError):
        ts.fetch_topk_data(torch.from_numpy(bad).cuda(), *stores)
    for s in stores:
        s.close()


def test_large_synthetic_dataset(tmp_path):
    # 200k items of ragged uint16 tokens: every gathered row checked against the host batch reader
    rng = np.random.Generator(np.random.PCG64(9))
    sizes = rng.integers(0, 130, size=200000)
    flat = rng.integers(0, 30522, size=int(sizes.sum())).astype(np.uint16)
    items = np.split(flat, np.cumsum(sizes)[:-1])
    prefix = str(tmp_path / "big")
    ts.write_mmap_dataset(prefix, items, dtype=np.uint16)
    ds = ts.MMapTokenDataset(prefix)
    store = ts.DeviceTokenStore(ds, device=0)
    ids = rng.integers(1, len(ds) + 1, size=(64, 32)).astype(np.int32)
    tok, lens = store.gather(torch.from_numpy(ids).cuda())
    tok, lens = tok.cpu().numpy().reshape(2048, -1), lens.cpu().numpy().ravel()
    want_flat, offs = ds.get_batch(ids.ravel() - 1)
    np.testing.assert_array_equal(lens, np.diff(offs))
    mask = np.arange(tok.shape[1])[None, :] < lens[:, None]
    np.testing.assert_array_equal(tok[mask], want_flat.astype(np.int32))
    assert (tok[~mask] == 0).all()
    store.close()


def test_gather_multi_equals_the_single_gathers():
    """arttok_gather_multi (one launch, four datasets) against arttok_gather per dataset: identical padded
    rows and lengths, with aligned and unaligned row widths, a pad value, and ids outside the datasets."""
    stores = [ts.DeviceTokenStore(ts.MMapTokenDataset(os.path.join(FX, n)), device=0) for n in NAMES]
    exp = np.load(os.path.join(FX, "expected.npz"))
    ids = exp["topkindex"].copy()
    ids[0, 1] = 100000      # outside every dataset: length -1, a row of pad
    ids[1, 0] = 0
    ids_dev = torch.from_numpy(ids).cuda()
    for max_lens, pads in (([None] * 4, None), ([64, 8, 61, 7], [0, -1, 3, 9]), ([4, 4, 4, 4], [5, 5, 5, 5])):
        multi = ts.gather_multi(stores, ids_dev, max_lens=max_lens, pads=pads)
        for j, store in enumerate(stores):
            want_t, want_l = store.gather(ids_dev, max_len=max_lens[j], pad=pads[j] if pads else 0)
            got_t, got_l = multi[j]
            assert torch.equal(got_t, want_t) and torch.equal(got_l, want_l), (NAMES[j], max_lens[j])
    assert int(multi[0][1][0, 1].item()) == -1 and int(multi[0][1][1, 0].item()) == -1
    # fewer than four stores, and the argument checks
    two = ts.gather_multi(stores[:2], ids_dev)
    assert torch.equal(two[1][0], stores[1].gather(ids_dev)[0])
    with pytest.raises(ValueError):
        ts.gather_multi(stores + stores[:1], ids_dev)
    for s in stores:
        s.close()
