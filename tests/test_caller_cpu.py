"""gloo tests (world 2 and 3, CPU) of the caller-side SPMD protocol in art_amd.caller: the query
gathers and the per-rank split mirror art_model.py:345-365 and evaluate.py:140-177.  The searcher is
the ORACLE here (no GPU in this container); tests/test_spmd_gpu.py drives the same functions with a
SpmdSearcher on the GPU box."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from art_amd.caller import gather_queries, spmd_evaluate_search, spmd_get_topk, varsize_gather_nograd
from oracle import mips_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _OracleSearcher(object):
    """Stands in for SpmdSearcher: every rank returns the exact global top-k."""

    def __init__(self, rows, ids):
        self.rows, self.ids, self.calls = rows, ids, 0

    def search(self, queries, top_k, score_dtype=torch.float16):
        self.calls += 1
        s, i, _ = orc.exact_topk(queries.numpy().astype(np.float64), self.rows, self.ids, top_k)
        return torch.from_numpy(s).to(score_dtype), torch.from_numpy(i)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = np.load(os.path.join(GOLDEN, "fx_small_fp16.npz"))
        rows = fx["rows"].astype(np.float64)
        queries = torch.from_numpy(fx["queries"].astype(np.float32))
        searcher = _OracleSearcher(rows, fx["ids"])
        # training form: the same number of queries on every rank
        b = queries.shape[0] // world
        mine = queries[rank * b:(rank + 1) * b]
        allq = gather_queries(mine)
        assert torch.equal(allq, queries[:world * b])
        d, t = spmd_get_topk(searcher, mine, 1)
        assert d.dtype == torch.float16 and t.dtype == torch.int32 and d.shape == (b, 1)
        np.save(os.path.join(out_dir, "train_i%d.npy" % rank), t.numpy())
        # evaluation form: ragged, one rank may have nothing
        sizes = [3, 0, 5][:world] if world == 3 else [5, 2]
        lo = sum(sizes[:rank])
        mine = queries[lo:lo + sizes[rank]]
        gathered, allsizes = varsize_gather_nograd(mine)
        assert allsizes == sizes and torch.equal(gathered, queries[:sum(sizes)])
        calls = searcher.calls
        d1, t1, _ = spmd_evaluate_search(searcher, mine, 32)
        assert searcher.calls == calls + 1
        d2, t2, _ = spmd_evaluate_search(searcher, mine, 32, shard_size=3)
        assert searcher.calls == calls + 1 + -(-sum(sizes) // 3)
        assert torch.equal(t1, t2) and torch.equal(d1, d2) and t1.shape == (sizes[rank], 32)
        np.save(os.path.join(out_dir, "eval_i%d.npy" % rank), t1.numpy())
        np.save(os.path.join(out_dir, "eval_lo%d.npy" % rank), np.asarray([lo, sizes[rank]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_caller_protocol_gloo(tmp_path, world):
    port = 29100 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fx = np.load(os.path.join(GOLDEN, "fx_small_fp16.npz"))
    b = fx["queries"].shape[0] // world
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("train_i%d.npy" % r)), fx["ids_k1"][r * b:(r + 1) * b])
        lo, n = np.load(tmp_path / ("eval_lo%d.npy" % r))
        np.testing.assert_array_equal(np.load(tmp_path / ("eval_i%d.npy" % r)), fx["ids_k32"][lo:lo + n])


def test_caller_protocol_single_process():
    fx = np.load(os.path.join(GOLDEN, "fx_small_fp16.npz"))
    searcher = _OracleSearcher(fx["rows"].astype(np.float64), fx["ids"])
    q = torch.from_numpy(fx["queries"].astype(np.float32))
    d, t = spmd_get_topk(searcher, q, 100)
    np.testing.assert_array_equal(t.numpy(), fx["ids_k100"])
    d, t, sizes = spmd_evaluate_search(searcher, q, 32, shard_size=16)
    assert sizes == [q.shape[0]]
    np.testing.assert_array_equal(t.numpy(), fx["ids_k32"])
