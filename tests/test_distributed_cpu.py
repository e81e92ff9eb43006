"""world_size-2 (and 3) gloo tests of the SPMD host logic on the CPU: shard bounds, the single packed
all-gather of (score, id) candidates, and the merge contract.  The per-rank searcher and the merge are
the ORACLE here (no GPU in this container); on the GPU box the same spmd_search drives MipsEngine
(tests/test_search_gpu.py::test_spmd_search_world1_nccl)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from art_amd.distributed import exchange_candidates, shard_bounds, spmd_search
from oracle import mips_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _worker(rank, world, port, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = np.load(os.path.join(GOLDEN, "fx_shard_edges.npz"))
        rows = fx["rows_i8"].astype(np.float64)
        queries = fx["queries_i8"].astype(np.float64)
        ids = fx["ids"]
        lo, hi = shard_bounds(rows.shape[0], world)[rank]

        def local_search(q, kk):
            s, i, _ = orc.exact_topk(q.numpy(), rows[lo:hi], ids[lo:hi], kk)
            pad = kk - s.shape[1]
            if pad > 0:
                s = np.concatenate([s, np.full((s.shape[0], pad), -np.inf, np.float32)], 1)
                i = np.concatenate([i, np.full((i.shape[0], pad), np.iinfo(np.int32).max, np.int32)], 1)
            return torch.from_numpy(s), torch.from_numpy(i)

        def merge(s, i):
            assert s.shape == (world, queries.shape[0], k) and s.dtype == torch.float32 and i.dtype == torch.int32
            ms, mi = orc.merge_topk(s.numpy(), i.numpy(), k)
            return torch.from_numpy(ms), torch.from_numpy(mi)

        s, i = spmd_search(local_search, merge, torch.from_numpy(queries), k)
        np.save(os.path.join(out_dir, "s%d.npy" % rank), s.numpy())
        np.save(os.path.join(out_dir, "i%d.npy" % rank), i.numpy())
        # the packed exchange returns every rank's block in rank order, bit for bit
        sc = torch.full((2, 3), float(rank) + 0.5)
        idt = torch.full((2, 3), rank, dtype=torch.int32)
        gs, gi = exchange_candidates(sc, idt)
        assert gs.shape == (world, 2, 3)
        for r in range(world):
            assert torch.all(gs[r] == r + 0.5) and torch.all(gi[r] == r)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 32), (3, 8)])
def test_spmd_search_gloo(tmp_path, world, k):
    port = 29500 + (os.getpid() % 400) + world
    mp.spawn(_worker, args=(world, port, k, str(tmp_path)), nprocs=world, join=True)
    fx = np.load(os.path.join(GOLDEN, "fx_shard_edges.npz"))
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("i%d.npy" % r)), fx["ids_k%d" % k])
        np.testing.assert_array_equal(np.load(tmp_path / ("s%d.npy" % r)), fx["scores_k%d" % k])


def test_shard_bounds_are_torch_chunk():
    assert shard_bounds(21015324, 8)[0] == (0, 2626916)
    assert shard_bounds(21015324, 8)[7] == (7 * 2626916, 21015324)
    assert shard_bounds(5, 8) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert shard_bounds(1003, 4) == orc.shard_bounds(1003, 4)


# ---- world 8: the collective protocol of SpmdSearcher (single exchange and pipelined chunks) with an
# overflow flag injected on ONE rank in ONE chunk: every rank must repeat the step and issue the same
# number of collectives (a rank that disagreed would hang the next all-gather on a real node)
class _FakeEngine(object):
    """Stands in for MipsEngine on the CPU: the oracle searches this rank's rows into the packed layout
    [Q*k score bits | Q*k ids | 4 status words] (include/artmips.h); merge_packed_sync is the oracle's
    merge plus the OR of the lists' status words, as merge_kernel does it."""

    def __init__(self, rows, ids, overflow_on_call):
        self.rows, self.ids = rows, ids
        self.overflow_on_call = overflow_on_call  # index of the search_local_async call that overflows
        self.calls = 0
        self.grown = 0

    def search_local_async(self, slot, queries, top_k, out_packed):
        q = queries.numpy().astype(np.float64)
        s, i, _ = orc.exact_topk(q, self.rows, self.ids, top_k)
        pad = top_k - s.shape[1]
        if pad > 0:
            s = np.concatenate([s, np.full((s.shape[0], pad), -np.inf, np.float32)], 1)
            i = np.concatenate([i, np.full((i.shape[0], pad), np.iinfo(np.int32).max, np.int32)], 1)
        plane = q.shape[0] * top_k
        out_packed[:plane] = torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).view(torch.int32).reshape(-1)
        out_packed[plane:2 * plane] = torch.from_numpy(np.ascontiguousarray(i, dtype=np.int32)).reshape(-1)
        out_packed[2 * plane:] = 0
        if self.calls == self.overflow_on_call and self.grown == 0:
            out_packed[2 * plane] = 1
        self.calls += 1

    @staticmethod
    def merge_packed_sync(packed, n_q, top_k, score_dtype=torch.float32):
        plane = n_q * top_k
        s = packed[:, :plane].contiguous().view(torch.float32).reshape(packed.shape[0], n_q, top_k).numpy()
        i = packed[:, plane:2 * plane].reshape(packed.shape[0], n_q, top_k).numpy()
        flag = 0
        for g in range(packed.shape[0]):
            flag |= int(packed[g, 2 * plane])
        ms, mi = orc.merge_topk(s, i, top_k)
        return torch.from_numpy(ms).to(score_dtype), torch.from_numpy(mi), flag

    def grow_lists(self):
        self.grown += 1


def _worker8(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd import distributed as D

        rng = np.random.Generator(np.random.PCG64(808))
        n, dim = 4000, 32
        rows = rng.integers(-4, 5, size=(n, dim)).astype(np.float64)
        ids = (rng.permutation(n) + 1).astype(np.int32)
        lo, hi = shard_bounds(n, world)[rank]
        gathers = [0]
        real_all_gather = dist.all_gather

        def counting_all_gather(*a, **kw):
            gathers[0] += 1
            return real_all_gather(*a, **kw)

        dist.all_gather = counting_all_gather
        report = {}
        # (a) one exchange per step: 70 queries x top-12, rank 5 overflows on its first search
        queries = rng.integers(-3, 4, size=(70, dim)).astype(np.float32)
        eng = _FakeEngine(rows[lo:hi], ids[lo:hi], overflow_on_call=0 if rank == 5 else -1)
        searcher = D.SpmdSearcher(eng)
        s, i = searcher.search(torch.from_numpy(queries), 12, torch.float32)
        want_s, want_i, _ = orc.exact_topk(queries.astype(np.float64), rows, ids, 12)
        assert np.array_equal(i.numpy(), want_i) and np.array_equal(s.numpy(), want_s)
        assert searcher.last_retries == 1 and eng.calls == 2 and eng.grown == 1
        report["single"] = gathers[0]
        # (b) pipelined: 700 queries in chunks of 256 (3 chunks), rank 2 overflows in chunk 1
        gathers[0] = 0
        queries = rng.integers(-3, 4, size=(700, dim)).astype(np.float32)
        eng = _FakeEngine(rows[lo:hi], ids[lo:hi], overflow_on_call=1 if rank == 2 else -1)
        searcher = D.SpmdSearcher(eng, chunk_queries=256)
        s, i = searcher.search(torch.from_numpy(queries), 9, torch.float32)
        want_s, want_i, _ = orc.exact_topk(queries.astype(np.float64), rows, ids, 9)
        assert np.array_equal(i.numpy(), want_i) and np.array_equal(s.numpy(), want_s)
        assert searcher.last_retries == 1 and eng.calls == 6 and eng.grown == 1
        report["pipelined"] = gathers[0]
        with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
            f.write("%d %d" % (report["single"], report["pipelined"]))
    finally:
        dist.destroy_process_group()


def test_spmd_searcher_world8_overflow_on_one_rank_keeps_collectives_in_step(tmp_path):
    world = 8
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    counts = [open(tmp_path / ("r%d.txt" % r)).read().split() for r in range(world)]
    # every rank: 2 all-gathers for the single-exchange step (first try + the repeat), 6 for the three
    # chunks (3 + 3); identical on all ranks
    assert all(c == ["2", "6"] for c in counts), counts
