"""CPU tests of the oracle itself: it must reproduce every committed golden vector, its two
functions must agree on tie-free data, and the merge oracle must equal a global top-k."""
import os

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_oracle_reproduces_fx_small_fp16(golden_dir):
    fx = _load(golden_dir, "fx_small_fp16.npz")
    rows, queries, ids = fx["rows"], fx["queries"], fx["ids"]
    assert rows.dtype == np.float16 and queries.dtype == np.float16
    for k in (1, 32, 100):
        s, i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, k)
        np.testing.assert_array_equal(i, fx["ids_k%d" % k])
        np.testing.assert_array_equal(s, fx["scores_k%d" % k])


@pytest.mark.parametrize("name,ks", [("fx_ties_int.npz", (1, 5, 32, 100)), ("fx_shard_edges.npz", (8, 32)),
                                     ("fx_k_eq_n.npz", (97,)), ("fx_q1.npz", (33,))])
def test_oracle_reproduces_integer_fixtures(golden_dir, name, ks):
    fx = _load(golden_dir, name)
    rows = fx["rows_i8"].astype(np.float64)
    queries = fx["queries_i8"].astype(np.float64)
    for k in ks:
        s, i, _ = orc.exact_topk(queries, rows, fx["ids"], k)
        np.testing.assert_array_equal(i, fx["ids_k%d" % k])
        np.testing.assert_array_equal(s, fx["scores_k%d" % k])


def test_ties_are_broken_by_lower_id(golden_dir):
    fx = _load(golden_dir, "fx_ties_int.npz")
    s, i = fx["scores_k100"], fx["ids_k100"].astype(np.int64)
    tie = np.diff(s, axis=1) == 0
    assert tie.sum() > 100, "fixture must contain many exact ties"
    assert np.all(np.diff(i, axis=1)[tie] > 0)


def test_restatement_agrees_with_exact_on_tie_free_data():
    # fp32 torch.matmul + torch.topk (the reference's algorithm, art_index.py:280,294) must pick the
    # same ids as the defined semantics when no two scores are within fp32 rounding of each other
    rng = np.random.Generator(np.random.PCG64(3))
    rows = rng.standard_normal((3000, 128)).astype(np.float32)
    queries = rng.standard_normal((5, 128)).astype(np.float32)
    ids = np.arange(1, 3001, dtype=np.int32)
    s, i, full = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, 20, extra=1)
    gaps = -np.diff(-np.sort(-full, axis=1)[:, :22], axis=1)
    assert gaps.min() > 1e-4
    d, ri = orc.reference_restatement(torch.from_numpy(queries), torch.from_numpy(rows), ids, 20)
    np.testing.assert_array_equal(ri.numpy(), i[:, :20])
    np.testing.assert_allclose(d.numpy(), s[:, :20], rtol=1e-5, atol=1e-5)
    # sharded restatement (torch.chunk slabs, art_index.py:251-253,285-291) gives the same
    d4, ri4 = orc.reference_restatement(torch.from_numpy(queries), torch.from_numpy(rows), ids, 20, n_shards=7)
    np.testing.assert_array_equal(ri4.numpy(), ri.numpy())


def test_shard_bounds_match_torch_chunk():
    for n in (1, 7, 16, 1003, 21015324):
        for g in (1, 2, 3, 4, 8):
            chunks = torch.chunk(torch.empty(n, 0), g, dim=0) if n < 10 ** 6 else None
            b = [hi - lo for lo, hi in orc.shard_bounds(n, g) if hi > lo]
            assert sum(b) == n
            if chunks is not None:
                assert b == [c.shape[0] for c in chunks]
    assert orc.shard_bounds(21015324, 8)[0] == (0, 2626916)


def test_merge_oracle_equals_global_topk(golden_dir):
    fx = _load(golden_dir, "fx_shard_edges.npz")
    rows = fx["rows_i8"].astype(np.float64)
    queries = fx["queries_i8"].astype(np.float64)
    ids = fx["ids"]
    k = 32
    for g in (1, 2, 4, 8):
        sl, il = [], []
        for lo, hi in orc.shard_bounds(rows.shape[0], g):
            s, i, _ = orc.exact_topk(queries, rows[lo:hi], ids[lo:hi], k)
            pad = k - s.shape[1]
            if pad > 0:
                s = np.concatenate([s, np.full((s.shape[0], pad), -np.inf, np.float32)], axis=1)
                i = np.concatenate([i, np.full((i.shape[0], pad), np.iinfo(np.int32).max, np.int32)], axis=1)
            sl.append(s)
            il.append(i)
        ms, mi = orc.merge_topk(np.stack(sl), np.stack(il), k)
        np.testing.assert_array_equal(mi, fx["ids_k32"])
        np.testing.assert_array_equal(ms, fx["scores_k32"])


def test_check_topk_accepts_oracle_and_rejects_errors(golden_dir):
    fx = _load(golden_dir, "fx_small_fp16.npz")
    rows = fx["rows"].astype(np.float64)
    queries = fx["queries"].astype(np.float64)
    ids = fx["ids"]
    s, i = fx["scores_k32"], fx["ids_k32"]
    assert orc.check_topk(s, i, queries, rows, ids, 32) == 0
    bad = i.copy()
    bad[0, 5] = bad[0, 6]
    with pytest.raises(AssertionError):
        orc.check_topk(s, bad, queries, rows, ids, 32)
    bad = i.copy()
    outside = next(v for v in ids if v not in set(i[2].tolist()))
    bad[2, -1] = outside
    with pytest.raises(AssertionError):
        orc.check_topk(s, bad, queries, rows, ids, 32)
