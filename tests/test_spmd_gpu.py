"""The SPMD path (one process per shard) end to end on the GPU box: two ranks share the one GPU,
each owns half of the rows in its own MipsEngine, candidates travel through torch.distributed
(gloo here: RCCL refuses two ranks on one device; the driver's multi-GPU bench uses "nccl"), and
every rank merges on the device.  Compared with the golden fixture and the oracle."""
import faulthandler
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _worker(rank, world, port, out_dir):
    faulthandler.dump_traceback_later(90, exit=True, file=sys.stderr)  # a stuck rank says where
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd.distributed import engine_spmd_search, shard_bounds
        from art_amd.engine import MipsEngine
        torch.cuda.set_device(0)
        fx = np.load(os.path.join(GOLDEN, "fx_shard_edges.npz"))
        rows = fx["rows_i8"].astype(np.float16)
        lo, hi = shard_bounds(rows.shape[0], world)[rank]
        eng = MipsEngine(768, torch.float16, [0])
        eng.build(rows[lo:hi], fx["ids"][lo:hi])
        q = torch.from_numpy(fx["queries_i8"].astype(np.float16)).cuda()
        for k in (8, 32):
            s, i = engine_spmd_search(eng, q, k, score_dtype=torch.float32)
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, "s%d_%d.npy" % (k, rank)), s.cpu().numpy())
            np.save(os.path.join(out_dir, "i%d_%d.npy" % (k, rank)), i.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_spmd_two_ranks_one_gpu(tmp_path, world):
    assert torch.cuda.is_available()
    port = 29600 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fx = np.load(os.path.join(GOLDEN, "fx_shard_edges.npz"))
    for k in (8, 32):
        for r in range(world):
            np.testing.assert_array_equal(np.load(tmp_path / ("i%d_%d.npy" % (k, r))), fx["ids_k%d" % k])
            np.testing.assert_array_equal(np.load(tmp_path / ("s%d_%d.npy" % (k, r))), fx["scores_k%d" % k])


def _caller_worker(rank, world, port, out_dir):
    faulthandler.dump_traceback_later(90, exit=True, file=sys.stderr)  # a stuck rank says where
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd.caller import spmd_evaluate_search, spmd_get_topk
        from art_amd.distributed import SpmdSearcher, shard_bounds
        from art_amd.engine import MipsEngine
        torch.cuda.set_device(0)
        fx = np.load(os.path.join(GOLDEN, "fx_small_fp16.npz"))
        lo, hi = shard_bounds(fx["rows"].shape[0], world)[rank]
        eng = MipsEngine(768, torch.float16, [0])
        eng.build(fx["rows"][lo:hi], fx["ids"][lo:hi])
        searcher = SpmdSearcher(eng)
        q = torch.from_numpy(fx["queries"]).cuda()
        b = q.shape[0] // world
        # art_model.py:345-365 without the broadcasts: every rank ends with ITS rows
        d, t = spmd_get_topk(searcher, q[rank * b:(rank + 1) * b], 32)
        assert d.dtype == torch.float16 and t.dtype == torch.int32 and d.is_cuda
        np.save(os.path.join(out_dir, "t%d.npy" % rank), t.cpu().numpy())
        np.save(os.path.join(out_dir, "d%d.npy" % rank), d.float().cpu().numpy())
        # evaluate.py:140-177 with ragged per-rank query counts, one call and chunked
        sizes = [5, 3] if world == 2 else [3, 0, 5]
        off = sum(sizes[:rank])
        d1, t1, allsizes = spmd_evaluate_search(searcher, q[off:off + sizes[rank]], 100)
        d2, t2, _ = spmd_evaluate_search(searcher, q[off:off + sizes[rank]], 100, shard_size=3)
        assert allsizes == sizes and torch.equal(t1, t2) and torch.equal(d1, d2)
        np.save(os.path.join(out_dir, "e%d.npy" % rank), t1.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_caller_protocol_spmd_on_gpu(tmp_path, world):
    port = 29200 + (os.getpid() % 300) + world
    mp.spawn(_caller_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fx = np.load(os.path.join(GOLDEN, "fx_small_fp16.npz"))
    b = fx["queries"].shape[0] // world
    sizes = [5, 3] if world == 2 else [3, 0, 5]
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("t%d.npy" % r)), fx["ids_k32"][r * b:(r + 1) * b])
        np.testing.assert_allclose(np.load(tmp_path / ("d%d.npy" % r)), fx["scores_k32"][r * b:(r + 1) * b],
                                   rtol=2e-3)  # fp16 output rounding
        off = sum(sizes[:r])
        np.testing.assert_array_equal(np.load(tmp_path / ("e%d.npy" % r)), fx["ids_k100"][off:off + sizes[r]])


def _pipelined_worker(rank, world, port, out_dir):
    faulthandler.dump_traceback_later(90, exit=True, file=sys.stderr)  # a stuck rank says where
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd.distributed import SpmdSearcher, shard_bounds
        from art_amd.engine import MipsEngine
        torch.cuda.set_device(0)
        rng = np.random.Generator(np.random.PCG64(77))
        rows = (rng.standard_normal((9001, 768)) * 0.05).astype(np.float16)
        ids = (rng.permutation(9001) + 1).astype(np.int32)
        queries = rng.standard_normal((700, 768)).astype(np.float16)
        lo, hi = shard_bounds(9001, world)[rank]
        eng = MipsEngine(768, torch.float16, [0])
        eng.build(rows[lo:hi], ids[lo:hi])
        q = torch.from_numpy(queries).cuda()
        # 700 queries in chunks of 256 (256 + 256 + 188): per-chunk exchange and merge ...
        s1, i1 = SpmdSearcher(eng, chunk_queries=256).search(q, 20, score_dtype=torch.float32)
        # ... against one exchange for the whole batch
        s2, i2 = SpmdSearcher(eng, chunk_queries=2048).search(q, 20, score_dtype=torch.float32)
        assert torch.equal(i1, i2) and torch.equal(s1, s2)
        np.save(os.path.join(out_dir, "pi_%d.npy" % rank), i1.cpu().numpy())
        np.save(os.path.join(out_dir, "ps_%d.npy" % rank), s1.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


def test_spmd_pipelined_large_batch(tmp_path):
    """The chunked large-batch step (exchange of chunk c behind the scan of chunk c+1) gives every
    rank the result of the one-exchange step and of the oracle."""
    from oracle import mips_oracle as orc
    world = 2
    port = 29000 + (os.getpid() % 300)
    mp.spawn(_pipelined_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.Generator(np.random.PCG64(77))
    rows = (rng.standard_normal((9001, 768)) * 0.05).astype(np.float16)
    ids = (rng.permutation(9001) + 1).astype(np.int32)
    queries = rng.standard_normal((700, 768)).astype(np.float16)
    i0 = np.load(tmp_path / "pi_0.npy")
    np.testing.assert_array_equal(i0, np.load(tmp_path / "pi_1.npy"))
    n_diff = orc.check_topk(np.load(tmp_path / "ps_0.npy"), i0, queries.astype(np.float64),
                            rows.astype(np.float64), ids, 20, rel_tol=1e-3)
    assert n_diff <= 28
