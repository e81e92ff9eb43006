"""CPU (gloo) tests of the SPMD index-refresh exchange: every rank encodes an interleaved slice of
the passages (like DistributedBatchSampler, samplers.py:142-148), commit() routes every row to the
rank that owns its id range."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from art_amd.distributed import shard_bounds
from art_amd.refresh import SpmdIndexWriter, route_by_owner


def test_route_by_owner():
    ids = torch.tensor([10, 1, 7, 4, 3, 9], dtype=torch.int64)
    rows = ids.float().unsqueeze(1).repeat(1, 2)
    i, r, counts = route_by_owner(ids, rows, 10, 1, 3)  # chunk 4: ids 1-4 | 5-8 | 9-10
    assert counts == [3, 1, 2]
    assert i.tolist() == [1, 4, 3, 7, 10, 9]
    assert torch.equal(r[:, 0], i.float())
    with pytest.raises(ValueError):
        route_by_owner(torch.tensor([11]), torch.zeros(1, 2), 10, 1, 3)


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = []
        w = SpmdIndexWriter(None, n, id_base=1, sink=lambda rows, ids, base: got.append((rows, ids, base)),
                            device=torch.device("cpu"))
        rng = np.random.Generator(np.random.PCG64(0))
        all_rows = rng.standard_normal((n, 8)).astype(np.float32)
        mine = np.arange(rank, n, world)  # interleaved slice
        for b in range(0, len(mine), 5):
            sel = mine[b:b + 5]
            w.add_block_data(torch.from_numpy(sel + 1), torch.from_numpy(all_rows[sel]))
        n_in = w.commit()
        lo, hi = shard_bounds(n, world)[rank]
        assert n_in == hi - lo
        rows, ids, base = got[0]
        assert base == 1 + lo
        order = torch.argsort(ids)
        assert ids[order].tolist() == list(range(lo + 1, hi + 1))
        np.testing.assert_array_equal(rows[order].numpy(), all_rows[lo:hi])
        np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 50)])
def test_spmd_writer_exchange_gloo(tmp_path, world, n):
    port = 29800 + (os.getpid() % 150) + world
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d.npy" % r)) for r in range(world))


# ---- streaming: several exchange rounds while batches keep arriving, ranks with different batch counts and
# sizes (one of them with nothing at all), a bounded buffer
def _stream_worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = []
        dim = 8
        flush = 40 * (dim * 4 + 8)   # a round every ~40 buffered rows
        w = SpmdIndexWriter(None, n, id_base=1, sink=lambda rows, ids, base: got.append((rows.clone(), ids.clone(), base)),
                            device=torch.device("cpu"), flush_bytes=flush)
        rng = np.random.Generator(np.random.PCG64(1))
        all_rows = rng.standard_normal((n, dim)).astype(np.float32)
        # consecutive ids are encoded at the same time, as with the reference's sampler: global batches of
        # 12 rows, split contiguously over the ranks that encode (the last rank encodes nothing)
        workers = world - 1
        batch = 12
        per = batch // workers
        for g0 in range(0, n, batch):
            if rank < workers:
                sel = np.arange(g0 + rank * per, min(g0 + (rank + 1) * per, n))
                sel = sel[sel < min(g0 + batch, n)]
                if len(sel):
                    w.add_block_data(torch.from_numpy(sel + 1), torch.from_numpy(all_rows[sel]))
        n_in = w.commit()
        lo, hi = shard_bounds(n, world)[rank]
        assert n_in == hi - lo, (n_in, lo, hi)
        ids = torch.cat([g[1] for g in got]) if got else torch.empty(0, dtype=torch.int64)
        rows = torch.cat([g[0] for g in got]) if got else torch.empty((0, dim))
        order = torch.argsort(ids)
        assert ids[order].tolist() == list(range(lo + 1, hi + 1))
        np.testing.assert_array_equal(rows[order].numpy(), all_rows[lo:hi])
        assert all(g[2] == 1 + lo for g in got)
        # what the rank held at any time: its accumulating batch (< flush + one batch) plus one round in
        # flight (what it sent + what the world sent to it: at most world x (flush + one batch) each way)
        one = flush + per * (dim * 4 + 8)
        assert w.peak_buffered_bytes <= one + (1 + world) * one, w.peak_buffered_bytes
        assert w.peak_buffered_bytes < n * (dim * 4 + 8) // 2, "the writer buffered like the commit-only form"
        with open(os.path.join(out_dir, "rounds%d.txt" % rank), "w") as f:
            f.write(str(w.rounds))
    finally:
        dist.destroy_process_group()


def test_spmd_writer_streams_in_rounds(tmp_path):
    world, n = 3, 1200
    port = 29700 + (os.getpid() % 90)
    mp.spawn(_stream_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    rounds = [int(open(tmp_path / ("rounds%d.txt" % r)).read()) for r in range(world)]
    assert len(set(rounds)) == 1 and rounds[0] >= 4, rounds   # identical on all ranks, and really several
