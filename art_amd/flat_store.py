"""Flat, mmap-able evidence store.

The reference's store is one pickle of a dict {id -> np.float16[dim]} (art_index.py:33-54,
63-100): loading 21M rows means unpickling 21M small arrays (minutes, >32 GB of Python
objects).  This file format holds the same information as three flat sections that can be
memory-mapped and handed to artmips_upload without a copy through Python objects:

    header   64 bytes  magic "ARTFLAT1", uint32 version, uint32 dtype (0 f16, 1 bf16, 2 f32),
                       uint64 n_rows, uint64 dim, zero padding
    ids      int32[n_rows], ascending, padded to a multiple of 64 bytes
    rows     dtype[n_rows, dim], row-major, in id order

Converters to and from the reference pickle keep the two formats interchangeable.
"""
import os
import struct

import numpy as np

ID_SENTINEL = -2139062144  # 0x80808080: the index marks id slots of rows not uploaded yet with it (artmips.h)

MAGIC = b"ARTFLAT1"
HEADER_BYTES = 64
_DT = {0: np.float16, 1: np.uint16, 2: np.float32}  # bf16 is carried as raw uint16


def _ids_bytes(n):
    return (n * 4 + 63) // 64 * 64


def write_flat(path, ids, rows, dtype_code=0):
    """Write ids (any integer array) and rows ([n, dim]; float16, float32, or uint16 holding
    bfloat16 bits) sorted by ascending id."""
    ids = np.asarray(ids)
    rows = np.asarray(rows)
    if rows.ndim != 2 or ids.shape != (rows.shape[0],):
        raise ValueError("ids must be [n] and rows [n, dim]")
    if rows.dtype != _DT[dtype_code]:
        raise ValueError("rows dtype %s does not match dtype code %d" % (rows.dtype, dtype_code))
    if ids.size and (ids.min() < -2 ** 31 or ids.max() > 2 ** 31 - 2):
        raise ValueError("passage ids must fit in int32")
    if np.any(ids == ID_SENTINEL):
        raise ValueError("passage id %d is reserved by the index (its 'row not uploaded yet' marker)" % ID_SENTINEL)
    order = np.argsort(ids, kind="stable")
    ids32 = ids[order].astype(np.int32)
    if np.any(np.diff(ids32) == 0):
        raise ValueError("Unexpectedly tried to overwrite block data")  # duplicate id
    n, dim = rows.shape
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC + struct.pack("<IIQQ", 1, dtype_code, n, dim).ljust(HEADER_BYTES - len(MAGIC), b"\0"))
        f.write(ids32.tobytes().ljust(_ids_bytes(n), b"\0"))
        # stream the rows in id order without materialising a second copy of the matrix
        step = 1 << 16
        for lo in range(0, n, step):
            f.write(np.ascontiguousarray(rows[order[lo:lo + step]]).tobytes())
    os.replace(tmp, path)


def save_engine_flat(engine, path, chunk_rows=1 << 17):
    """Write the live index of a MipsEngine as a flat store file, streaming it off the GPUs chunk by
    chunk (artmips_download): the on-disk artefact after a device-side refresh, where the reference
    has its merged pickle (indexer.py:156-163).  Safe to run on a background thread beside searches;
    raises ArtmipsError if the index is replaced meanwhile.  Returns the number of rows written.

    An index in id order (what the device refresh builds) is streamed straight through; any other
    row order (an index built from a store's dict) is placed by id through a memory map."""
    gen = engine.generation()
    n = int(engine.n_rows)
    dtype_code = {"torch.float16": 0, "torch.bfloat16": 1, "torch.float32": 2}[str(engine.storage_dtype)]
    np_dt = _DT[dtype_code]
    dim = engine.dim
    # pass 1: the ids (4 bytes a row) decide where every row goes
    ids = np.empty(n, dtype=np.int32)
    sorted_already = True
    prev = None
    tmp = path + ".tmp"
    rows_off = HEADER_BYTES + _ids_bytes(n)
    with open(tmp, "wb") as f:
        f.write(MAGIC + struct.pack("<IIQQ", 1, dtype_code, n, dim).ljust(HEADER_BYTES - len(MAGIC), b"\0"))
        f.write(b"\0" * _ids_bytes(n))
        for lo in range(0, n, chunk_rows):
            cnt = min(chunk_rows, n - lo)
            rows, cid = engine.download(lo, cnt, gen)
            ids[lo:lo + cnt] = cid
            if sorted_already:
                if (prev is not None and cid[0] <= prev) or (cnt > 1 and np.any(np.diff(cid) <= 0)):
                    sorted_already = False
                prev = int(cid[-1])
            f.write(rows.tobytes())  # in index order for now
    if not sorted_already:
        order = np.argsort(ids, kind="stable")
        if np.any(np.diff(ids[order]) == 0):
            os.remove(tmp)
            raise ValueError("Unexpectedly tried to overwrite block data")  # duplicate id
        rank = np.empty(n, dtype=np.int64)
        rank[order] = np.arange(n, dtype=np.int64)
        # pass 2: rows to their place in id order, chunk by chunk through a second file
        tmp2 = path + ".tmp2"
        src = np.memmap(tmp, dtype=np_dt, mode="r", offset=rows_off, shape=(n, dim))
        with open(tmp2, "wb") as f:
            f.write(MAGIC + struct.pack("<IIQQ", 1, dtype_code, n, dim).ljust(HEADER_BYTES - len(MAGIC), b"\0"))
            f.write(ids[order].tobytes().ljust(_ids_bytes(n), b"\0"))
            for lo in range(0, n, chunk_rows):
                f.write(np.ascontiguousarray(src[order[lo:lo + chunk_rows]]).tobytes())
        del src
        os.remove(tmp)
        os.replace(tmp2, path)
    else:
        with open(tmp, "r+b") as f:
            f.seek(HEADER_BYTES)
            f.write(ids.tobytes())
        os.replace(tmp, path)
    return n


def save_engine_flat_shared(engine, path, row0, n_total, rank, barrier, chunk_rows=1 << 17):
    """One flat store file written by several processes, each holding the rows [row0, row0 + its
    engine's rows) of an index that is in global id order (what the SPMD device refresh builds:
    art_amd.refresh.SpmdIndexWriter).  Collective: every rank calls it with the same path and
    n_total; `barrier` is a callable that returns once every rank has reached it
    (torch.distributed.barrier).  Rank 0 lays the file out, every rank streams its shard off its GPU
    into its slice through a memory map, rank 0 checks that the ids ascend and puts the file in place.
    The ranks must see one filesystem (they are the GPUs of one node).  Returns n_total."""
    dtype_code = {"torch.float16": 0, "torch.bfloat16": 1, "torch.float32": 2}[str(engine.storage_dtype)]
    np_dt = _DT[dtype_code]
    dim = engine.dim
    n_local = int(engine.n_rows)
    tmp = path + ".tmp"
    rows_off = HEADER_BYTES + _ids_bytes(n_total)
    total_bytes = rows_off + n_total * dim * np.dtype(np_dt).itemsize
    if rank == 0:
        with open(tmp, "wb") as f:
            f.write(MAGIC + struct.pack("<IIQQ", 1, dtype_code, n_total, dim).ljust(HEADER_BYTES - len(MAGIC), b"\0"))
            f.truncate(total_bytes)
    barrier()
    gen = engine.generation()
    ids_mm = np.memmap(tmp, dtype=np.int32, mode="r+", offset=HEADER_BYTES, shape=(n_total,))
    rows_mm = np.memmap(tmp, dtype=np_dt, mode="r+", offset=rows_off, shape=(n_total, dim))
    for lo in range(0, n_local, chunk_rows):
        cnt = min(chunk_rows, n_local - lo)
        rows, ids = engine.download(lo, cnt, gen)
        rows_mm[row0 + lo:row0 + lo + cnt] = rows
        ids_mm[row0 + lo:row0 + lo + cnt] = ids
    rows_mm.flush()
    ids_mm.flush()
    del rows_mm, ids_mm
    barrier()
    err = None
    if rank == 0:
        all_ids = np.memmap(tmp, dtype=np.int32, mode="r", offset=HEADER_BYTES, shape=(n_total,))
        if n_total > 1 and not bool(np.all(np.diff(all_ids) > 0)):
            err = ValueError("the shards are not in ascending id order: a shared flat file needs the "
                             "index the device refresh builds")
        del all_ids
        if err is None:
            os.replace(tmp, path)
        else:
            os.remove(tmp)
    barrier()
    if err is not None:
        raise err
    return n_total


class FlatStore(object):
    """Memory-mapped view of a flat store file: .ids int32 [n], .rows [n, dim], .dtype_code."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if len(head) < HEADER_BYTES or head[:8] != MAGIC:
            raise ValueError("%s is not an ARTFLAT1 file" % path)
        version, self.dtype_code, self.n_rows, self.dim = struct.unpack("<IIQQ", head[8:32])
        if version != 1 or self.dtype_code not in _DT:
            raise ValueError("unsupported flat store version %d / dtype %d" % (version, self.dtype_code))
        want = HEADER_BYTES + _ids_bytes(self.n_rows) + self.n_rows * self.dim * np.dtype(_DT[self.dtype_code]).itemsize
        if os.path.getsize(path) != want:
            raise ValueError("%s is truncated: %d bytes, expected %d" % (path, os.path.getsize(path), want))
        self.ids = np.memmap(path, dtype=np.int32, mode="r", offset=HEADER_BYTES, shape=(self.n_rows,))
        self.rows = np.memmap(path, dtype=_DT[self.dtype_code], mode="r",
                              offset=HEADER_BYTES + _ids_bytes(self.n_rows), shape=(self.n_rows, self.dim))

    def __len__(self):
        return self.n_rows

    def iter_chunks(self, chunk_rows=1 << 16):
        for lo in range(0, self.n_rows, chunk_rows):
            yield np.asarray(self.ids[lo:lo + chunk_rows]), np.asarray(self.rows[lo:lo + chunk_rows])


def store_to_flat(store, path):
    """Reference-format store (dict {id -> float16[dim]}) -> flat file."""
    ids = np.fromiter((int(k) for k in store.embed_data.keys()), dtype=np.int64, count=len(store.embed_data))
    rows = np.asarray(list(store.embed_data.values()), dtype=np.float16)
    write_flat(path, ids, rows, 0)


def flat_to_store(flat, embedding_path, rank=0):
    """Flat file -> reference-format store object (keys numpy.int64, values float16[dim] like
    art_index.py:61); call .save_shard()/.merge_shards_and_save() or pickle its state() to get the
    reference's file."""
    from .art_index import OpenRetreivalDataStore

    if flat.dtype_code != 0:
        raise ValueError("the reference pickle holds float16 rows")
    st = OpenRetreivalDataStore(embedding_path, load_from_path=False, rank=rank)
    for ids, rows in flat.iter_chunks():
        st.add_block_data(ids.astype(np.int64), rows)
    return st


__all__ = ["write_flat", "FlatStore", "store_to_flat", "flat_to_store", "save_engine_flat", "save_engine_flat_shared"]
