"""GPU tests of the drop-in classes (art_amd.art_index) used the way the reference's callers use
them: art_model.py:325-365 (build from a store, search, update_index) and evaluate.py:145-163
(chunked search with --shard-size queries per call)."""
import os

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc

pytestmark = pytest.mark.gpu


def _store(tmp_path, n=3000, dim=768, seed=3, ranks=2):
    from art_amd.art_index import OpenRetreivalDataStore
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
    ids = rng.permutation(n).astype(np.int64) + 1  # 1-based evidence ids, arbitrary order
    path = str(tmp_path / "evidence.pkl")
    # the indexer's protocol (indexer.py:146-165): every rank adds its batches, saves its shard,
    # rank 0 merges
    stores = []
    for r in range(ranks):
        st = OpenRetreivalDataStore(path, load_from_path=False, rank=r)
        sl = slice(r, n, ranks)
        for b in range(0, len(ids[sl]), 128):
            st.add_block_data(ids[sl][b:b + 128], rows[sl][b:b + 128])
        st.save_shard()
        stores.append(st)
    stores[0].merge_shards_and_save()
    assert len(stores[0].embed_data) == n
    return path, rows.astype(np.float16), ids


def test_build_search_update(tmp_path):
    from art_amd.art_index import DistributedBruteForceIndex, OpenRetreivalDataStore
    assert torch.cuda.is_available()
    path, rows, ids = _store(tmp_path)
    store = OpenRetreivalDataStore(path, load_from_path=True)
    index = DistributedBruteForceIndex(embed_size=768, embed_data=store)
    assert index.ngpu == torch.cuda.device_count()
    assert store.embed_data == {}  # add_embed_data clears the store (art_index.py:262)
    assert index.num_rows == 3000 and len(index.indices_arr) == 3000 and index.id_map[0] == index.indices_arr[0]
    rng = np.random.Generator(np.random.PCG64(8))
    queries = rng.standard_normal((64, 768)).astype(np.float16)
    q = torch.from_numpy(queries).cuda()
    dist, idx = index.search_mips_index(q, top_k=33, reconstruct=False)  # topk+1 of art_model.py:295
    assert dist.dtype == torch.float16 and idx.dtype == torch.int32
    assert dist.device == q.device and tuple(dist.shape) == (64, 33) and tuple(idx.shape) == (64, 33)
    # the caller broadcasts into buffers of exactly these dtypes (art_model.py:358-362)
    torch.empty(64, 33, dtype=torch.float16, device="cuda").copy_(dist)
    index.score_dtype = torch.float32
    d32, i32 = index.search_mips_index(q, top_k=33)
    assert torch.equal(idx, i32) and torch.equal(dist, d32.half())
    orc.check_topk(d32.cpu().numpy(), i32.cpu().numpy(), queries.astype(np.float64), rows.astype(np.float64),
                   ids, 33, rel_tol=1e-3)
    # evaluate.py:149-160: the same queries in chunks of --shard-size 16 give the same rows
    parts = [index.search_mips_index(q[i:i + 16], top_k=33, reconstruct=False)[1] for i in range(0, 64, 16)]
    assert torch.equal(torch.cat(parts, 0), i32)

    # refresh: the indexer rewrites the pickle, update_index() reloads it (art_model.py:336-342)
    st = OpenRetreivalDataStore(path, load_from_path=True)
    first = next(iter(st.embed_data))
    st.embed_data[first] = np.float16(queries[0] * 4)  # plant an obvious winner for query 0
    import pickle
    with open(path, "wb") as f:
        pickle.dump(st.state(), f)
    index.update_index()
    d2, i2 = index.search_mips_index(q, top_k=5)
    assert int(i2[0, 0]) == int(first)
    index.reset_index()
    d3, i3 = index.search_mips_index(q, top_k=5)
    assert torch.equal(i2, i3)


def test_index_without_store_and_errors(tmp_path):
    from art_amd.art_index import DistributedBruteForceIndex
    index = DistributedBruteForceIndex(embed_size=768, embed_data=None)
    with pytest.raises(RuntimeError):
        index.search_mips_index(torch.zeros(1, 768, dtype=torch.float16, device="cuda"), 1)


def test_empty_query_batch_returns_empty_like_the_reference(tmp_path):
    """torch.matmul + torch.topk on a [0, 768] query batch give [0, k] results (art_index.py:280,294);
    the drop-in does the same instead of raising."""
    from art_amd.art_index import DistributedBruteForceIndex, OpenRetreivalDataStore
    path, rows, ids = _store(tmp_path, n=300)
    index = DistributedBruteForceIndex(embed_size=768, embed_data=OpenRetreivalDataStore(path))
    d, i = index.search_mips_index(torch.zeros(0, 768, dtype=torch.float16, device="cuda"), 7, reconstruct=False)
    assert tuple(d.shape) == (0, 7) and tuple(i.shape) == (0, 7)
    assert d.dtype == torch.float16 and i.dtype == torch.int32 and d.is_cuda


@pytest.mark.parametrize("use_gpu", [False, True])
def test_faiss_shaped_index(tmp_path, use_gpu):
    """FaissMIPSIndex keeps the reference's surface (art_index.py:103-197): constructor, numpy
    returns (float32 distances, int64 ids), reconstruct=True returning the stored vectors, and
    update_index / reset_index.  use_gpu=False is the exact-fp32 IndexFlatIP (fp32 queries are
    not rounded), use_gpu=True the useFloat16 GPU index (queries rounded to fp16)."""
    from art_amd.art_index import FaissMIPSIndex, OpenRetreivalDataStore
    path, rows, ids = _store(tmp_path, n=2500, seed=5)
    store = OpenRetreivalDataStore(path, load_from_path=True)
    index = FaissMIPSIndex(embed_size=768, embed_data=store, use_gpu=use_gpu)
    assert store.embed_data == {}  # art_index.py:175
    rng = np.random.Generator(np.random.PCG64(21))
    queries = rng.standard_normal((10, 768)).astype(np.float32)
    q = torch.from_numpy(queries)  # a CPU tensor, as evaluate.py may pass (art_index.py:188 calls .cpu())
    dist, idx = index.search_mips_index(q, 20, reconstruct=False)
    assert isinstance(dist, np.ndarray) and isinstance(idx, np.ndarray)
    assert dist.dtype == np.float32 and idx.dtype == np.int64 and dist.shape == (10, 20) and idx.shape == (10, 20)
    # evaluate.py:155-157 turns them into the tensors it broadcasts
    torch.from_numpy(dist).half().cuda(), torch.from_numpy(idx).int().cuda()
    q_seen = queries.astype(np.float16) if use_gpu else queries
    orc.check_topk(dist, idx.astype(np.int32), q_seen.astype(np.float64), rows.astype(np.float64), ids, 20,
                   rel_tol=1e-3 if use_gpu else 1e-5)
    # reconstruct=True (the default): faiss's search_and_reconstruct triple
    d2, i2, vec = index.search_mips_index(q.cuda(), 20)
    assert np.array_equal(i2, idx) and np.array_equal(d2, dist)
    assert vec.dtype == np.float32 and vec.shape == (10, 20, 768)
    row_of = {int(p): r for r, p in enumerate(ids)}
    want = np.stack([[rows[row_of[int(p)]].astype(np.float32) for p in line] for line in idx])
    assert np.array_equal(vec, want)
    # the returned vectors reproduce the returned distances
    redo = np.einsum("qd,qkd->qk", q_seen.astype(np.float64), vec.astype(np.float64))
    assert np.allclose(redo, dist, rtol=1e-3, atol=1e-4)
    index.update_index()
    d3, i3 = index.search_mips_index(q, 20, reconstruct=False)
    assert np.array_equal(i3, idx)
    index.reset_index()
    d4, i4 = index.search_mips_index(q, 20, reconstruct=False)
    assert np.array_equal(i4, idx)
