"""Drop-in mirror of the reference's evidence store and brute-force MIPS index.

Same names, arguments and error behaviour as
``/root/reference/megatron/data/art_index.py`` (``detach`` :12,
``OpenRetreivalDataStore`` :16-100, ``DistributedBruteForceIndex`` :200-304), so
that ``megatron/model/art_model.py:12``, ``megatron/indexer.py:7`` and
``tasks/.../evaluate.py:15`` work after a one-line import swap.
The search itself runs in libartmips.so (hand-written gfx950 kernels behind the
C ABI of include/artmips.h); there is no torch.matmul / torch.topk fallback.

Differences from the reference, all deliberate:
  * ties are broken by the LOWER passage id and scores are ranked in fp32 before
    they are rounded to fp16 for the caller (the reference ranks fp16-rounded
    scores with torch.topk's unspecified tie order, art_index.py:280-294);
  * one process still drives every visible GPU (art_index.py:204), but only
    Q*k (score, id) pairs per GPU move between devices instead of [Q, N/G] slabs;
  * ``FaissMIPSIndex`` keeps the reference's constructor, methods and numpy return types
    (art_index.py:103-197) but is served by the same HIP library: faiss itself is out of
    scope and is never imported.
"""
import os
import pickle
import shutil

import numpy as np
import torch

from .flat_store import ID_SENTINEL


def _megatron():
    """(get_args, mpu) when this runs inside ART, else (None, None)."""
    try:
        from megatron import get_args, mpu  # type: ignore

        return get_args, mpu
    except Exception:
        return None, None


def _is_main_printer():
    _, mpu = _megatron()
    if mpu is None:
        return True
    try:
        return mpu.is_unitialized() or mpu.get_data_parallel_rank() == 0
    except Exception:
        return True


def detach(tensor):
    """Same contract as art_index.py:12-13: a host numpy copy of a tensor."""
    return tensor.detach().cpu().numpy()


class OpenRetreivalDataStore(object):
    """Evidence embeddings keyed by passage id, serialisable shard by shard.

    Mirrors art_index.py:16-100.  ``embed_data`` is a ``dict {id -> np.float16[dim]}``
    in insertion order; the on-disk artefact is one pickle ``{'embed_data': dict}``,
    byte-compatible with the reference (tests/golden/fx_store_pickle).
    """

    def __init__(self, embedding_path=None, load_from_path=True, rank=None):
        self.embed_data = dict()
        if embedding_path is None:
            get_args, _ = _megatron()
            if get_args is None:
                raise ValueError("embedding_path is required outside a megatron run")
            args = get_args()
            embedding_path = args.embedding_path
            rank = args.rank
        self.embedding_path = embedding_path
        self.rank = rank

        if load_from_path:
            self.load_from_file()

        self.temp_dir_name = os.path.splitext(self.embedding_path)[0] + '_tmp'

    def state(self):
        return {'embed_data': self.embed_data}

    def clear(self):
        """Drop the embeddings (the index owns them once they are on the GPUs)."""
        self.embed_data = dict()

    def load_from_file(self):
        if _is_main_printer():
            print("\n> Unpickling BlockData: ", self.embedding_path, flush=True)
        with open(self.embedding_path, 'rb') as f:
            state_dict = pickle.load(f)
        if _is_main_printer():
            print(">> Finished unpickling BlockData\n", flush=True)
        self.embed_data = state_dict['embed_data']

    def add_block_data(self, row_id, block_embeds, allow_overwrite=False):
        """Insert one batch; every row is cast to float16 (art_index.py:56-61).
        Raises ValueError on a duplicate id unless allow_overwrite."""
        for idx, embed in zip(row_id, block_embeds):
            if not allow_overwrite and idx in self.embed_data:
                raise ValueError("Unexpectedly tried to overwrite block data")
            self.embed_data[idx] = np.float16(embed)

    def save_shard(self):
        """Write this rank's embeddings to <embedding_path stem>_tmp/<rank>.pkl."""
        os.makedirs(self.temp_dir_name, exist_ok=True)
        with open('{}/{}.pkl'.format(self.temp_dir_name, self.rank), 'wb') as writer:
            pickle.dump(self.state(), writer)

    def merge_shards_and_save(self):
        """Fold every other rank's shard into this store (own shard first, the rest
        in os.listdir order), assert there is no id overlap, write the merged pickle
        and remove the temporary directory (art_index.py:72-100)."""
        shard_names = os.listdir(self.temp_dir_name)
        seen_own_shard = False

        for fname in shard_names:
            shard_rank = int(os.path.splitext(fname)[0])
            if shard_rank == self.rank:
                seen_own_shard = True
                continue
            with open(os.path.join(self.temp_dir_name, fname), 'rb') as f:
                data = pickle.load(f)
            before = len(self.embed_data)
            incoming = len(data['embed_data'])
            self.embed_data.update(data['embed_data'])
            assert len(self.embed_data) == before + incoming

        assert seen_own_shard

        with open(self.embedding_path, 'wb') as final_file:
            pickle.dump(self.state(), final_file)
        shutil.rmtree(self.temp_dir_name, ignore_errors=True)

        print("Finished merging {} shards for a total of {} embeds".format(
            len(shard_names), len(self.embed_data)), flush=True)

    # ---- not in the reference: bulk export used by the index upload ----------------------------
    def iter_chunks(self, chunk_rows=65536):
        """Yield (ids int64 [n], rows float16 [n, dim]) in dict insertion order."""
        ids, rows = [], []
        for idx, embed in self.embed_data.items():
            ids.append(idx)
            rows.append(embed)
            if len(ids) == chunk_rows:
                yield np.asarray(ids, dtype=np.int64), np.asarray(rows, dtype=np.float16)
                ids, rows = [], []
        if ids:
            yield np.asarray(ids, dtype=np.int64), np.asarray(rows, dtype=np.float16)


class _IdMap(object):
    """Read-only stand-in for the reference's ``id_map`` dict (row position -> passage id,
    art_index.py:257-259) that does not materialise 21M Python ints."""

    def __init__(self, indices_arr):
        self._arr = indices_arr

    def __getitem__(self, pos):
        return self._arr[pos]

    def __len__(self):
        return len(self._arr)

    def __contains__(self, pos):
        return 0 <= pos < len(self._arr)

    def items(self):
        return enumerate(self._arr)


class DistributedBruteForceIndex(object):
    """Exact MIPS index over every visible GPU; mirrors art_index.py:200-304."""

    def __init__(self, embed_size, embed_data=None, devices=None, storage_dtype=torch.float16):
        self.embed_size = embed_size
        self.embed_data = embed_data
        self.devices = list(devices) if devices is not None else list(range(torch.cuda.device_count()))
        self.ngpu = len(self.devices)
        self.storage_dtype = storage_dtype
        # fp16 is what the callers' broadcast buffers expect (art_model.py:358, evaluate.py:166)
        self.score_dtype = torch.float16

        self.evidence_embeds = None  # the MipsEngine that owns the device shards
        self.indices_arr = None
        self.id_map = dict()
        self.num_rows = 0
        self.chunksize = 0
        self.last_chunksize = 0
        self._set_mips_index()

    def _set_mips_index(self):
        if _is_main_printer():
            print("\n> Building Brute Force MIPS index", flush=True)
        if self.embed_data is not None:
            self.add_embed_data(self.embed_data)

    def reset_index(self):
        """Drop the device shards and rebuild from a fresh store object."""
        self._free()
        if self.embed_data is not None:
            embed_data_path = self.embed_data.embedding_path
            del self.embed_data
            self.embed_data = OpenRetreivalDataStore(embed_data_path)
        self._set_mips_index()

    def update_index(self):
        """Drop the device shards, reload the store from disk and rebuild."""
        self._free()
        if self.embed_data is not None:
            self.embed_data.load_from_file()
        self._set_mips_index()

    def _free(self):
        if self.evidence_embeds is not None:
            self.evidence_embeds.close()
        self.evidence_embeds = None

    def add_embed_data(self, all_embed_data):
        """Move the store's rows to the GPUs (contiguous ceil-div row chunks, art_index.py:251-255)
        and remember row position -> passage id.  Clears the store (art_index.py:262)."""
        from .engine import MipsEngine

        if self.ngpu < 1:
            raise RuntimeError("DistributedBruteForceIndex needs at least one GPU")
        n = len(all_embed_data.embed_data)
        if n < 1:
            raise ValueError("the store holds no embeddings")
        engine = MipsEngine(self.embed_size, self.storage_dtype, self.devices)
        engine.begin(n)
        all_ids = np.empty(n, dtype=np.int64)
        pos = 0
        for ids, rows in all_embed_data.iter_chunks():
            if rows.ndim != 2 or rows.shape[1] != self.embed_size:
                raise ValueError("store rows have shape %s, index expects [*, %d]"
                                 % (rows.shape, self.embed_size))
            if ids.min() < -2 ** 31 or ids.max() > 2 ** 31 - 2:
                raise ValueError("passage ids must fit in int32")
            if np.any(ids == ID_SENTINEL):
                raise ValueError("passage id %d is reserved by the index (its 'row not uploaded yet' marker)"
                                 % ID_SENTINEL)
            all_ids[pos:pos + len(ids)] = ids
            pos += len(ids)
            if self.storage_dtype == torch.float16:
                engine.upload_host(rows, ids.astype(np.int32))
            else:
                engine.upload_host(torch.from_numpy(rows).to(self.storage_dtype), ids.astype(np.int32))
        engine.finalize()
        self._free()
        self.evidence_embeds = engine
        self.num_rows = n
        self.chunksize = -(-n // self.ngpu)
        self.last_chunksize = max(0, n - self.chunksize * (self.ngpu - 1))  # trailing shards may be empty (N < G)
        self.indices_arr = all_ids
        self.id_map = _IdMap(all_ids)

        all_embed_data.clear()

        if _is_main_printer():
            print(">>> Finished adding block data to GPU index", flush=True)

    def add_flat_store(self, flat):
        """Build the index from an art_amd.flat_store.FlatStore: rows go from the page cache to the
        GPUs in 64K-row chunks, no Python object per row (not in the reference)."""
        from .engine import MipsEngine

        want = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[self.storage_dtype]
        if flat.dtype_code != want or flat.dim != self.embed_size:
            raise ValueError("flat store is dtype %d dim %d, index expects dtype %d dim %d"
                             % (flat.dtype_code, flat.dim, want, self.embed_size))
        engine = MipsEngine(self.embed_size, self.storage_dtype, self.devices)
        engine.begin(flat.n_rows)
        for ids, rows in flat.iter_chunks():
            if self.storage_dtype == torch.bfloat16:
                engine.upload_host(torch.from_numpy(rows.view(np.int16)).view(torch.bfloat16), ids)
            else:
                engine.upload_host(rows, ids)
        engine.finalize()
        self._free()
        self.evidence_embeds = engine
        self.num_rows = flat.n_rows
        self.chunksize = -(-flat.n_rows // self.ngpu)
        self.last_chunksize = max(0, flat.n_rows - self.chunksize * (self.ngpu - 1))
        self.indices_arr = np.asarray(flat.ids).astype(np.int64)
        self.id_map = _IdMap(self.indices_arr)

    def save_flat(self, path, background=False):
        """Write the live index to `path` as an art_amd.flat_store file (streamed off the GPUs, ids
        ascending).  With background=True the export runs on a thread beside the searches and the
        Thread object is returned (join it before the process exits; a refresh that replaces the
        index meanwhile makes the export fail cleanly, its error is kept in thread.error).  Not in
        the reference, whose index has no way back to the disk: its checkpoint is the pickle the
        indexer wrote (indexer.py:156-163), which the device refresh path never writes."""
        from .flat_store import save_engine_flat

        if self.evidence_embeds is None:
            raise RuntimeError("MIPS index is empty: nothing to save")
        engine = self.evidence_embeds
        if not background:
            return save_engine_flat(engine, path)
        import threading

        def run():
            try:
                t.rows_written = save_engine_flat(engine, path)
            except Exception as e:  # noqa: BLE001 - reported through the thread object
                t.error = e

        t = threading.Thread(target=run, name="artmips-save-flat", daemon=False)
        t.error = None
        t.rows_written = 0
        t.start()
        return t

    def to_store(self, embedding_path=None, rank=0):
        """The live index as a reference-format store object (dict {numpy.int64 id -> float16[dim]}
        in row order), e.g. to write the reference's pickle with save_shard()/merge_shards_and_save()
        or pickle.dump(store.state()).  One Python object per row: meant for modest sizes and for
        compatibility, the flat file is the fast artefact."""
        if self.evidence_embeds is None:
            raise RuntimeError("MIPS index is empty: nothing to export")
        if self.storage_dtype != torch.float16:
            raise ValueError("the reference store holds float16 rows")
        engine = self.evidence_embeds
        path = embedding_path
        if path is None and self.embed_data is not None:
            path = self.embed_data.embedding_path
        store = OpenRetreivalDataStore(path, load_from_path=False, rank=rank)
        gen = engine.generation()
        step = 1 << 16
        for lo in range(0, int(engine.n_rows), step):
            rows, ids = engine.download(lo, min(step, int(engine.n_rows) - lo), gen)
            store.add_block_data(ids.astype(np.int64), rows)
        return store

    def search_mips_index(self, query_embeds, top_k, reconstruct=True):
        """Top-k passages by inner product.

        Returns (distances [Q, k] float16, indices [Q, k] int32) on the query's device, rows
        sorted by descending score, ties by lower passage id.  ``reconstruct`` is accepted and
        ignored exactly as in the reference (art_index.py:267-270)."""
        if self.evidence_embeds is None:
            raise RuntimeError("MIPS index is empty: call add_embed_data first")
        if query_embeds.dim() == 2 and query_embeds.shape[0] == 0:
            # no queries: the reference's matmul + topk give empty [0, k] results
            return (torch.empty((0, int(top_k)), dtype=self.score_dtype, device=query_embeds.device),
                    torch.empty((0, int(top_k)), dtype=torch.int32, device=query_embeds.device))
        return self.evidence_embeds.search(query_embeds, top_k, score_dtype=self.score_dtype)


class FaissMIPSIndex(object):
    """The reference's faiss-backed sibling (art_index.py:103-197) with the same constructor,
    methods and return types, served by libartmips instead of faiss.

    ``search_mips_index`` takes a torch tensor on any device and returns **numpy** arrays like
    ``faiss.IndexIDMap.search``: ``(distances float32 [Q, k], ids int64 [Q, k])``; with
    ``reconstruct=True`` (the reference's default, art_index.py:190-193:
    ``search_and_reconstruct``) the stored vectors come back as a third array float32 [Q, k, dim].
    The callers already accept this form (evaluate.py:155-157).

    ``use_gpu=False`` is the reference's exact fp32 ``IndexFlatIP``: rows are kept in fp32 (the
    store's fp16 values widened, as art_index.py:171 does) and fp32 queries are not rounded.
    ``use_gpu=True`` is its ``useFloat16`` GPU index (art_index.py:125-131): fp16 rows, queries
    rounded to fp16, fp32 accumulation.  Both run on the GPUs; there is no CPU search path.
    Differences: exact ties are broken by the lower passage id (faiss: unspecified), and slots beyond
    the number of rows cannot occur because top_k <= rows is enforced (faiss pads with id -1)."""

    def __init__(self, embed_size, embed_data=None, use_gpu=False, devices=None):
        self.embed_size = embed_size
        self.embed_data = embed_data
        self.use_gpu = use_gpu
        self.devices = devices

        self.mips_index = None
        self._row_of_id = None
        self._set_mips_index()

    def _set_mips_index(self):
        if _is_main_printer():
            print("\n> Building FAISS-shaped index on libartmips", flush=True)
        storage = torch.float16 if self.use_gpu else torch.float32
        self.mips_index = DistributedBruteForceIndex(self.embed_size, None, devices=self.devices,
                                                     storage_dtype=storage)
        self.mips_index.score_dtype = torch.float32
        if self.embed_data is not None:
            self.add_embed_data(self.embed_data)

    def _drop(self):
        if self.mips_index is not None:
            self.mips_index._free()
        self.mips_index = None

    def reset_index(self):
        self._drop()
        if self.embed_data is not None:
            embed_data_path = self.embed_data.embedding_path
            del self.embed_data
            self.embed_data = OpenRetreivalDataStore(embed_data_path)
        self._set_mips_index()

    def update_index(self):
        self._drop()
        if self.embed_data is not None:
            self.embed_data.load_from_file()
        self._set_mips_index()

    def add_embed_data(self, all_embed_data):
        """Add every embedding of the store to the index and clear the store (art_index.py:164-180)."""
        self.mips_index.add_embed_data(all_embed_data)
        self._row_of_id = None

    def _rows_of_ids(self, ids):
        # passage id -> row position, built on first use (reconstruct only)
        if self._row_of_id is None:
            arr = self.mips_index.indices_arr
            order = np.argsort(arr, kind="stable")
            self._row_of_id = (arr[order], order)
        sorted_ids, order = self._row_of_id
        return order[np.searchsorted(sorted_ids, ids)]

    def search_mips_index(self, query_embeds, top_k, reconstruct=True):
        index = self.mips_index
        if index is None or index.evidence_embeds is None:
            raise RuntimeError("MIPS index is empty: call add_embed_data first")
        dev = torch.device("cuda", index.devices[0])
        # faiss takes float32 queries (art_index.py:188); the library rounds them to the storage dtype
        q = query_embeds.detach().to(device=dev, dtype=torch.float32)
        distances, ids = index.search_mips_index(q, top_k, reconstruct=False)
        distances = distances.cpu().numpy()
        ids = ids.cpu().numpy().astype(np.int64)
        if not reconstruct:
            return distances, ids
        rows = index.evidence_embeds.fetch_rows(self._rows_of_ids(ids.reshape(-1)))
        return distances, ids, rows.reshape(ids.shape[0], ids.shape[1], self.embed_size)


__all__ = ["detach", "OpenRetreivalDataStore", "DistributedBruteForceIndex", "FaissMIPSIndex"]
