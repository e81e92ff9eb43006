#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ (run in the build container).

Search fixtures come from oracle/mips_oracle.py (the reference ships no vectors for
this path and its search cannot run without CUDA: SURVEY.md §8c).  The store
fixture comes from the REFERENCE'S OWN OpenRetreivalDataStore, imported by path
from /root/reference with a stub `megatron` module: that pins the pickle format,
dtype casts, key types and merge order.

Every fixture is an .npz of inputs + expected outputs; nothing of the reference's
source or bytecode is written anywhere.
"""
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mips_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
D = 768


def min_rel_gap(scores_sorted):
    """smallest relative gap between adjacent entries of a descending float64 score row"""
    d = -np.diff(scores_sorted)
    return float(np.min(d / np.maximum(np.abs(scores_sorted[:-1]), 1.0)))


def gen_random(seed_rows, seed_q, n, q, dim, dtype, scale=0.05):
    rng = np.random.Generator(np.random.PCG64(seed_rows))
    rows = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    rngq = np.random.Generator(np.random.PCG64(seed_q))
    queries = rngq.standard_normal((q, dim)).astype(np.float32)
    return rows, queries


def fx_small_fp16():
    # gap-checked: regenerate until every adjacent gap inside the top-(k+1) exceeds 2e-5 relative,
    # far above fp32 accumulation error (~1e-6 * sum|a*b|), so ids are order-independent
    seed = 1234
    while True:
        rows32, q32 = gen_random(seed, 4321, 2048, 8, D, "f16")
        rows = rows32.astype(np.float16)
        queries = q32.astype(np.float16)
        ids = np.arange(1, rows.shape[0] + 1, dtype=np.int32)
        s, i, full = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, 100, extra=1)
        full_sorted = -np.sort(-full, axis=1)[:, :102]
        gap = min(min_rel_gap(r) for r in full_sorted)
        if gap > 2e-5:
            break
        seed += 1
    out = {"rows": rows, "queries": queries, "ids": ids, "seed": np.int64(seed), "min_rel_gap": np.float64(gap)}
    for k in (1, 32, 100):
        out["scores_k%d" % k] = s[:, :k]
        out["ids_k%d" % k] = i[:, :k]
    np.savez(os.path.join(OUT, "fx_small_fp16.npz"), **out)
    print("fx_small_fp16: seed", seed, "min rel gap", gap)


def fx_ties_int():
    # integer-valued embeddings in [-8, 8]: every dot product is an integer < 2^24, exact in fp32
    # under any summation order, so exact ties are real ties.  Rows are duplicated on purpose and
    # ids are a permutation, so the expected order inside a tie is by LOWER ID, not by row position.
    rng = np.random.Generator(np.random.PCG64(77))
    base = rng.integers(-8, 9, size=(250, D)).astype(np.int8)
    rows = np.concatenate([base, base, base[:100], rng.integers(-8, 9, size=(400, D)).astype(np.int8)])
    perm = rng.permutation(rows.shape[0])
    rows = rows[perm]
    ids = (rng.permutation(rows.shape[0]) + 1).astype(np.int32)
    queries = rng.integers(-8, 9, size=(16, D)).astype(np.int8)
    # make ties at the very top for some queries: query = one of the duplicated rows
    queries[0] = base[3]
    queries[1] = base[17]
    out = {"rows_i8": rows, "queries_i8": queries, "ids": ids}
    for k in (1, 5, 32, 100):
        s, i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, k)
        out["scores_k%d" % k] = s
        out["ids_k%d" % k] = i
    np.savez_compressed(os.path.join(OUT, "fx_ties_int.npz"), **out)
    n_tied = int(np.sum(np.diff(out["scores_k100"], axis=1) == 0))
    print("fx_ties_int: rows", rows.shape, "tied adjacent pairs in top-100:", n_tied)


def fx_shard_edges():
    # N = 1003 is not divisible by 2, 4 or 8.  Winners are planted on the first/last row of every
    # shard for G in {2,4,8} and inside the short last shard; the expected global result does not
    # depend on G.
    n = 1003
    rng = np.random.Generator(np.random.PCG64(99))
    rows = rng.integers(-3, 4, size=(n, D)).astype(np.int8)
    queries = rng.integers(-3, 4, size=(4, D)).astype(np.int8)
    edges = set()
    for g in (2, 4, 8):
        for lo, hi in orc.shard_bounds(n, g):
            if hi > lo:
                edges.update([lo, hi - 1])
    edges.update([n - 1, n - 2, n - 3])
    edges = sorted(edges)
    for j, r in enumerate(edges):
        # a scaled copy of query (j % 4) scores far above the noise floor; the scale varies so
        # that the planted rows have distinct scores
        qi = j % 4
        rows[r] = np.clip(queries[qi].astype(np.int32) * (2 + (j % 3)), -8, 8).astype(np.int8)
    ids = np.arange(1, n + 1, dtype=np.int32)
    out = {"rows_i8": rows, "queries_i8": queries, "ids": ids, "edges": np.asarray(edges, dtype=np.int32)}
    for k in (8, 32):
        s, i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, k)
        out["scores_k%d" % k] = s
        out["ids_k%d" % k] = i
    np.savez_compressed(os.path.join(OUT, "fx_shard_edges.npz"), **out)
    print("fx_shard_edges: planted", len(edges), "rows")


def fx_k_eq_n_and_q1():
    rng = np.random.Generator(np.random.PCG64(5))
    rows = rng.integers(-8, 9, size=(97, D)).astype(np.int8)
    queries = rng.integers(-8, 9, size=(3, D)).astype(np.int8)
    ids = (rng.permutation(97) * 7 + 11).astype(np.int32)
    s, i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, 97)
    np.savez_compressed(os.path.join(OUT, "fx_k_eq_n.npz"), rows_i8=rows, queries_i8=queries, ids=ids,
                        scores_k97=s, ids_k97=i)
    rows = rng.integers(-8, 9, size=(700, D)).astype(np.int8)
    queries = rng.integers(-8, 9, size=(1, D)).astype(np.int8)
    ids = np.arange(1, 701, dtype=np.int32)
    s, i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, 33)
    np.savez_compressed(os.path.join(OUT, "fx_q1.npz"), rows_i8=rows, queries_i8=queries, ids=ids,
                        scores_k33=s, ids_k33=i)
    print("fx_k_eq_n, fx_q1 written")


def fx_bf16_vs_fp32():
    # C5's tolerance check in miniature: the same fp32 embeddings stored as bf16 and searched must
    # give scores within 1e-3 relative of the fp32-storage exact scores *of the bf16-rounded data*,
    # and ids identical wherever the fp32 gap exceeds the bf16 rounding bound.
    rows32, q32 = gen_random(2024, 4242, 1536, 8, D, "f32")
    ids = np.arange(1, rows32.shape[0] + 1, dtype=np.int32)
    rb = orc.to_storage(rows32, "bf16")
    qb = orc.to_storage(q32, "bf16")
    s_bf, i_bf, _ = orc.exact_topk(qb, rb, ids, 50)
    s_32, i_32, _ = orc.exact_topk(q32.astype(np.float64), rows32.astype(np.float64), ids, 50)
    np.savez(os.path.join(OUT, "fx_bf16_vs_fp32.npz"), rows_f32=rows32, queries_f32=q32, ids=ids,
             scores_bf16_k50=s_bf, ids_bf16_k50=i_bf, scores_f32_k50=s_32, ids_f32_k50=i_32)
    print("fx_bf16_vs_fp32: id agreement bf16 vs fp32 storage:", float(np.mean(i_bf == i_32)))


def fx_store_pickle():
    """Run the reference's own OpenRetreivalDataStore (by-path import, stub megatron) through
    add_block_data -> save_shard (3 ranks) -> merge_shards_and_save and keep the merged pickle
    plus the inputs that produced it."""
    ref_file = "/root/reference/megatron/data/art_index.py"
    if not os.path.exists(ref_file):
        print("fx_store_pickle: reference not present, skipped")
        return
    stub = types.ModuleType("megatron")
    mpu = types.ModuleType("megatron.mpu")
    mpu.is_unitialized = lambda: True
    mpu.get_data_parallel_rank = lambda: 0
    stub.mpu = mpu
    stub.get_args = lambda: None
    saved = {k: sys.modules.get(k) for k in ("megatron", "megatron.mpu")}
    sys.modules["megatron"] = stub
    sys.modules["megatron.mpu"] = mpu
    try:
        spec = importlib.util.spec_from_file_location("ref_art_index", ref_file)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    rng = np.random.Generator(np.random.PCG64(31))
    tmp = tempfile.mkdtemp()
    try:
        path = os.path.join(tmp, "evidence.pkl")
        batches = {}
        stores = []
        for rank in range(3):
            st = ref.OpenRetreivalDataStore(path, load_from_path=False, rank=rank)
            # indexer.py:146-149: row_id numpy int64 [b], block_embeds numpy fp16/fp32 [b, 768]
            ids = np.arange(1 + rank, 64 + 1, 3, dtype=np.int64)
            emb = rng.standard_normal((len(ids), D)).astype(np.float32 if rank != 1 else np.float16)
            st.add_block_data(ids, emb)
            st.save_shard()
            batches[rank] = (ids, emb)
            stores.append(st)
        listdir_order = os.listdir(stores[0].temp_dir_name)
        stores[0].merge_shards_and_save()
        merged_keys = np.asarray(list(stores[0].embed_data.keys()), dtype=np.int64)
        merged_rows = np.asarray(list(stores[0].embed_data.values()), dtype=np.float16)
        key_type = type(next(iter(stores[0].embed_data.keys()))).__name__
        val = next(iter(stores[0].embed_data.values()))
        dst = os.path.join(OUT, "fx_store_pickle")
        os.makedirs(dst, exist_ok=True)
        shutil.copy(path, os.path.join(dst, "merged.pkl"))
        np.savez(os.path.join(dst, "inputs.npz"),
                 ids0=batches[0][0], emb0=batches[0][1], ids1=batches[1][0], emb1=batches[1][1],
                 ids2=batches[2][0], emb2=batches[2][1], merged_keys=merged_keys, merged_rows=merged_rows)
        meta = {"listdir_order": listdir_order, "key_type": key_type, "value_dtype": str(val.dtype),
                "value_shape": list(val.shape), "tmp_dir_removed": not os.path.exists(stores[0].temp_dir_name),
                "pickle_bytes": os.path.getsize(path)}
        # the duplicate-id error of the reference
        try:
            stores[0].add_block_data(np.asarray([1], dtype=np.int64), np.zeros((1, D), dtype=np.float32))
            meta["duplicate_error"] = None
        except ValueError as e:
            meta["duplicate_error"] = str(e)
        with open(os.path.join(dst, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("fx_store_pickle:", meta)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    fx_small_fp16()
    fx_ties_int()
    fx_shard_edges()
    fx_k_eq_n_and_q1()
    fx_bf16_vs_fp32()
    fx_store_pickle()
