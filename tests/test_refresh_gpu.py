"""GPU tests of the index-refresh upload path: device batches scattered by passage id into the next
generation (no pickle round trip), atomic swap, integrity errors, flat-store upload, and the SPMD
exchange with two ranks sharing the GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mips_oracle as orc

pytestmark = pytest.mark.gpu


def _data(n=5000, dim=768, seed=4):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
    queries = rng.standard_normal((9, dim)).astype(np.float16)
    return rows, queries


@pytest.mark.parametrize("n_shards", [1, 3])
def test_device_writer_equals_store_path(tmp_path, n_shards):
    from art_amd.art_index import DistributedBruteForceIndex, OpenRetreivalDataStore
    from art_amd.refresh import DeviceIndexWriter
    assert torch.cuda.is_available()
    rows, queries = _data()
    n = rows.shape[0]
    ids = np.arange(1, n + 1, dtype=np.int64)
    # reference protocol: store -> index
    store = OpenRetreivalDataStore(str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    store.add_block_data(ids, rows)
    ref_index = DistributedBruteForceIndex(768, store, devices=[0] * n_shards)
    ref_index.score_dtype = torch.float32
    q = torch.from_numpy(queries).cuda()
    s_ref, i_ref = ref_index.search_mips_index(q, 40)

    # refresh protocol: fp32 encoder output on the GPU, batches of 128 in shuffled order
    index = DistributedBruteForceIndex(768, None, devices=[0] * n_shards)
    index.score_dtype = torch.float32
    w = DeviceIndexWriter(index, n, id_base=1)
    perm = np.random.Generator(np.random.PCG64(0)).permutation(n)
    d_rows = torch.from_numpy(rows).cuda()
    for b in range(0, n, 128):
        sel = torch.from_numpy(perm[b:b + 128]).cuda()
        w.add_block_data(sel + 1, d_rows[sel])
    w.commit()
    assert index.num_rows == n and index.indices_arr[0] == 1 and index.indices_arr[-1] == n
    s, i = index.search_mips_index(q, 40)
    assert torch.equal(i, i_ref) and torch.equal(s, s_ref)  # same fp16 rounding, same kernels
    orc.check_topk(s.cpu().numpy(), i.cpu().numpy(), queries.astype(np.float64),
                   rows.astype(np.float16).astype(np.float64), ids, 40, rel_tol=1e-3)

    # second refresh on the live index: the old generation keeps answering until commit
    rows2 = rows.copy()
    rows2[17] = queries[0].astype(np.float32) * 3  # id 18 becomes the clear winner for query 0
    w2 = DeviceIndexWriter(index, n, id_base=1)
    w2.add_block_data(torch.arange(1, n // 2 + 1).cuda(), torch.from_numpy(rows2[:n // 2]).cuda())
    s_mid, i_mid = index.search_mips_index(q, 40)
    assert torch.equal(i_mid, i_ref)
    w2.add_block_data(np.arange(n // 2 + 1, n + 1), rows2[n // 2:])  # numpy batches work too
    w2.commit()
    s2, i2 = index.search_mips_index(q, 40)
    assert int(i2[0, 0]) == 18


def test_refresh_integrity_errors():
    from art_amd._lib import ArtmipsError
    from art_amd.art_index import DistributedBruteForceIndex
    from art_amd.refresh import DeviceIndexWriter
    rows, _ = _data(n=300)
    d_rows = torch.from_numpy(rows).cuda()
    index = DistributedBruteForceIndex(768, None, devices=[0, 0])
    w = DeviceIndexWriter(index, 300)
    w.add_block_data(torch.arange(1, 301).cuda(), d_rows)
    w.add_block_data(torch.tensor([5]).cuda(), d_rows[:1])  # id 5 twice
    with pytest.raises(ValueError) as e:
        w.commit()
    assert "overwrite block data" in str(e.value)
    w = DeviceIndexWriter(index, 300)
    w.add_block_data(torch.arange(1, 300).cuda(), d_rows[:299])  # id 300 never arrives
    with pytest.raises(ArtmipsError) as e:
        w.commit()
    assert "never uploaded" in str(e.value)


def test_flat_store_upload(tmp_path):
    from art_amd.art_index import DistributedBruteForceIndex
    from art_amd.flat_store import FlatStore, write_flat
    rows, queries = _data(n=3000)
    ids = (np.random.Generator(np.random.PCG64(2)).permutation(3000) + 1).astype(np.int64)
    path = str(tmp_path / "e.flat")
    write_flat(path, ids, rows.astype(np.float16))
    index = DistributedBruteForceIndex(768, None, devices=[0, 0])
    index.add_flat_store(FlatStore(path))
    index.score_dtype = torch.float32
    s, i = index.search_mips_index(torch.from_numpy(queries).cuda(), 25)
    orc.check_topk(s.cpu().numpy(), i.cpu().numpy(), queries.astype(np.float64),
                   rows.astype(np.float16).astype(np.float64), ids, 25, rel_tol=1e-3)
    assert np.all(np.diff(index.indices_arr) > 0)


def _spmd_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd.distributed import engine_spmd_search
        from art_amd.engine import MipsEngine
        from art_amd.refresh import SpmdIndexWriter
        torch.cuda.set_device(0)
        rows, queries = _data(n=4001)
        n = rows.shape[0]
        eng = MipsEngine(768, torch.float16, [0])
        w = SpmdIndexWriter(eng, n, id_base=1)
        mine = np.arange(rank, n, world)
        for b in range(0, len(mine), 128):
            sel = mine[b:b + 128]
            w.add_block_data(torch.from_numpy(sel + 1).cuda(), torch.from_numpy(rows[sel]).cuda())
        w.commit(save_flat=os.path.join(out_dir, "spmd.flat"))  # every rank writes its shard's slice
        s, i = engine_spmd_search(eng, torch.from_numpy(queries).cuda(), 30, score_dtype=torch.float32)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "s%d.npy" % rank), s.cpu().numpy())
        np.save(os.path.join(out_dir, "i%d.npy" % rank), i.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


def test_spmd_writer_two_ranks_one_gpu(tmp_path):
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_spmd_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rows, queries = _data(n=4001)
    ids = np.arange(1, 4002, dtype=np.int64)
    for r in range(2):
        s, i = np.load(tmp_path / ("s%d.npy" % r)), np.load(tmp_path / ("i%d.npy" % r))
        orc.check_topk(s, i, queries.astype(np.float64), rows.astype(np.float16).astype(np.float64), ids, 30,
                       rel_tol=1e-3)
    # the one flat file the two ranks wrote together holds the whole index in id order
    from art_amd.flat_store import FlatStore
    flat = FlatStore(str(tmp_path / "spmd.flat"))
    assert flat.n_rows == 4001 and np.array_equal(flat.ids, ids.astype(np.int32))
    assert np.array_equal(np.array(flat.rows), rows.astype(np.float16))
    assert not os.path.exists(str(tmp_path / "spmd.flat.tmp"))


def _stream_worker(rank, world, port, out_dir, plant_duplicate):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from art_amd.distributed import engine_spmd_search
        from art_amd.engine import MipsEngine
        from art_amd.refresh import SpmdIndexWriter
        torch.cuda.set_device(0)
        rows, queries = _data(n=12001)
        n = rows.shape[0]
        eng = MipsEngine(768, torch.float16, [0])
        batch = 128
        row_bytes = 768 * 4 + 8
        flush = 3 * batch * row_bytes    # a round every 3 batches of this rank
        w = SpmdIndexWriter(eng, n, id_base=1, flush_bytes=flush)
        # the reference's sampler: global batches of consecutive ids, split contiguously over the ranks
        n_batches = 0
        for g0 in range(0, n, batch * world):
            sel = np.arange(g0 + rank * batch, min(g0 + (rank + 1) * batch, n))
            if len(sel) == 0:
                continue
            n_batches += 1
            if plant_duplicate and rank == 1 and n_batches == 20:
                sel = sel.copy()
                sel[3] = 17           # passage 18 again, from another rank, in a LATER round than its first copy
            w.add_block_data(torch.from_numpy(sel + 1).cuda(), torch.from_numpy(rows[sel]).cuda())
        assert w.rounds >= 3, "the batches of this test are meant to trigger several rounds before commit"
        outcome = "ok"
        try:
            w.commit()
        except ValueError as e:
            outcome = "dup:" + str(e)
        except Exception as e:  # noqa: BLE001 (the missing-row error of the rank that lost a row to the duplicate)
            outcome = "err:" + str(e)
        # bounded buffer: own batch + one round in flight (own send + what both ranks sent here)
        one = flush + batch * row_bytes
        assert w.peak_buffered_bytes <= one * (2 + world), (w.peak_buffered_bytes, one)
        assert w.peak_buffered_bytes < (n // world) * row_bytes // 2
        with open(os.path.join(out_dir, "out%d.txt" % rank), "w") as f:
            f.write("%s\n%d" % (outcome, w.rounds))
        if not plant_duplicate:
            s, i = engine_spmd_search(eng, torch.from_numpy(queries).cuda(), 30, score_dtype=torch.float32)
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, "s%d.npy" % rank), s.cpu().numpy())
            np.save(os.path.join(out_dir, "i%d.npy" % rank), i.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


def test_spmd_writer_streams_in_rounds_two_ranks_one_gpu(tmp_path):
    """Streaming refresh (indexer.py:146-165 without the filesystem): >= 3 exchange rounds before commit,
    a bounded buffer, and the index that comes out answers like the oracle."""
    port = 29600 + (os.getpid() % 90)
    mp.spawn(_stream_worker, args=(2, port, str(tmp_path), False), nprocs=2, join=True)
    rows, queries = _data(n=12001)
    ids = np.arange(1, 12002, dtype=np.int64)
    outs = [open(tmp_path / ("out%d.txt" % r)).read().split("\n") for r in range(2)]
    assert outs[0][0] == "ok" and outs[1][0] == "ok", outs
    assert outs[0][1] == outs[1][1] and int(outs[0][1]) >= 10, outs   # same number of rounds on both ranks
    for r in range(2):
        s, i = np.load(tmp_path / ("s%d.npy" % r)), np.load(tmp_path / ("i%d.npy" % r))
        orc.check_topk(s, i, queries.astype(np.float64), rows.astype(np.float16).astype(np.float64), ids, 30,
                       rel_tol=1e-3)


def test_spmd_writer_streaming_detects_a_duplicate_in_a_later_round(tmp_path):
    """Passage 18 arrives in round 1 from rank 0 and again, many rounds later, from rank 1 in place of
    another passage: the owner of passage 18 must fail its commit with the reference's duplicate error
    (art_index.py:59), whatever round the copies came in."""
    port = 29500 + (os.getpid() % 90)
    mp.spawn(_stream_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    outs = [open(tmp_path / ("out%d.txt" % r)).read().split("\n") for r in range(2)]
    assert outs[0][0].startswith("dup:") and "overwrite block data" in outs[0][0], outs   # rank 0 owns id 18
    assert outs[0][1] == outs[1][1], outs


@pytest.mark.parametrize("storage", [torch.float16, torch.bfloat16])
def test_export_after_device_refresh_roundtrip(tmp_path, storage):
    """The way back to disk after a device-side refresh (the reference's index checkpoint is the pickle
    its indexer writes, indexer.py:156-163): commit(save_flat=...) streams the new generation into a
    flat file on a background thread while searches go on; the file reloads into an identical index."""
    from art_amd.art_index import DistributedBruteForceIndex
    from art_amd.flat_store import FlatStore
    from art_amd.refresh import DeviceIndexWriter
    rows, queries = _data(n=6001)
    n = rows.shape[0]
    index = DistributedBruteForceIndex(768, None, devices=[0, 0, 0], storage_dtype=storage)
    index.score_dtype = torch.float32
    w = DeviceIndexWriter(index, n, id_base=1)
    perm = np.random.Generator(np.random.PCG64(3)).permutation(n)
    d_rows = torch.from_numpy(rows).cuda()
    for b in range(0, n, 500):
        sel = torch.from_numpy(perm[b:b + 500]).cuda()
        w.add_block_data(sel + 1, d_rows[sel])
    path = str(tmp_path / "evidence.flat")
    saver = w.commit(save_flat=path)          # background thread
    q = torch.from_numpy(queries).cuda().to(storage)
    s0, i0 = index.search_mips_index(q, 25)   # searches run beside the export
    saver.join(60)
    assert not saver.is_alive() and saver.error is None and saver.rows_written == n
    flat = FlatStore(path)
    assert flat.n_rows == n and flat.dim == 768 and np.array_equal(flat.ids, np.arange(1, n + 1))
    want = torch.from_numpy(rows).to(storage)
    got = torch.from_numpy(np.array(flat.rows))
    got = got.view(torch.bfloat16) if storage == torch.bfloat16 else got
    assert torch.equal(got, want)
    again = DistributedBruteForceIndex(768, None, devices=[0], storage_dtype=storage)
    again.score_dtype = torch.float32
    again.add_flat_store(flat)
    s1, i1 = again.search_mips_index(q, 25)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)


def test_export_unsorted_index_and_generation_guard(tmp_path):
    """An index built from a store keeps the dict's insertion order (art_index.py:244); its export is
    placed by id.  to_store() gives the reference-format dict back.  A download that started before a
    refresh fails instead of mixing two generations."""
    from art_amd._lib import ArtmipsError
    from art_amd.art_index import DistributedBruteForceIndex, OpenRetreivalDataStore
    from art_amd.flat_store import FlatStore
    rows, queries = _data(n=3000, seed=9)
    n = rows.shape[0]
    ids = np.random.Generator(np.random.PCG64(5)).permutation(n).astype(np.int64) + 1
    store = OpenRetreivalDataStore(str(tmp_path / "e.pkl"), load_from_path=False, rank=0)
    store.add_block_data(ids, rows)
    index = DistributedBruteForceIndex(768, store, devices=[0, 0])
    assert index.save_flat(str(tmp_path / "u.flat")) == n
    flat = FlatStore(str(tmp_path / "u.flat"))
    assert np.array_equal(flat.ids, np.arange(1, n + 1))
    order = np.argsort(ids)
    assert np.array_equal(np.asarray(flat.rows), rows.astype(np.float16)[order])
    # the reference-format store, in the index's row order, bit for bit
    back = index.to_store()
    assert list(back.embed_data.keys()) == list(ids) and isinstance(next(iter(back.embed_data)), np.int64)
    first = back.embed_data[ids[0]]
    assert first.dtype == np.float16 and np.array_equal(first, rows[0].astype(np.float16))
    assert back.embedding_path == str(tmp_path / "e.pkl")
    # generation guard
    eng = index.evidence_embeds
    gen = eng.generation()
    r, i = eng.download(10, 5, gen)
    assert np.array_equal(i, ids[10:15].astype(np.int32)) and np.array_equal(r, rows[10:15].astype(np.float16))
    eng.begin(n)
    eng.upload_host(rows.astype(np.float16), ids.astype(np.int32))
    eng.finalize()
    assert eng.generation() == gen + 1
    with pytest.raises(ArtmipsError):
        eng.download(10, 5, gen)
    with pytest.raises(ArtmipsError):
        eng.download(n - 2, 5)  # beyond the last row


def test_close_during_background_export_is_safe(tmp_path):
    """update_index() / close() right after save_flat(background=True): the handle is not destroyed
    under a running download chunk, and the export ends with an error instead of a crash."""
    from art_amd.art_index import DistributedBruteForceIndex
    from art_amd.refresh import DeviceIndexWriter
    rows, _ = _data(n=20000)
    index = DistributedBruteForceIndex(768, None, devices=[0])
    w = DeviceIndexWriter(index, rows.shape[0], id_base=1)
    w.add_block_data(torch.arange(1, rows.shape[0] + 1).cuda(), torch.from_numpy(rows).cuda())
    w.commit()
    import art_amd.flat_store as fs
    saver = None
    orig = fs.save_engine_flat

    def slow_save(engine, path, chunk_rows=256):  # many small chunks, so that close() lands between two
        return orig(engine, path, chunk_rows)

    fs.save_engine_flat = slow_save
    try:
        saver = index.save_flat(str(tmp_path / "x.flat"), background=True)
        index._free()  # what update_index() / reset_index() do first
        saver.join(60)
    finally:
        fs.save_engine_flat = orig
    assert not saver.is_alive()
    # either the export had finished already, or it stopped with the "closed" error; never a crash
    assert saver.error is None or "closed" in str(saver.error)
