"""MipsEngine: thin Python owner of one libartmips handle.

PyTorch is plumbing here (device memory for queries/results and the current HIP
stream); all search work happens in the HIP library behind the C ABI.  The
reference-shaped classes in ``art_amd.art_index`` are built on top of this.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import ArtmipsError, check, lib

_TORCH2ABI = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}
_NP2ABI = {np.dtype(np.float16): _lib.F16, np.dtype(np.float32): _lib.F32}


def _abi_dtype(dt):
    if dt in _TORCH2ABI:
        return _TORCH2ABI[dt]
    raise ValueError("unsupported dtype %r (need float16, bfloat16 or float32)" % (dt,))


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device(object):
    """torch.cuda.device(dev) without its cost when dev is the current device already (the per-step
    path: the GPU idles while the host gets from one step's result to the next step's first launch)."""
    __slots__ = ("dev", "ctx")

    def __init__(self, dev):
        self.dev = dev
        self.ctx = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class MipsEngine(object):
    """Exact MIPS top-k over row shards resident on one or more GPUs.

    devices: list of CUDA(HIP) device indices, one shard per entry (the same index
    may appear twice).  Defaults to every visible device, as the reference does
    (art_index.py:204).
    """

    def __init__(self, dim, storage_dtype=torch.float16, devices=None):
        if storage_dtype not in (torch.float16, torch.bfloat16, torch.float32):
            raise ValueError("storage dtype must be float16, bfloat16 or float32")
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        if len(devices) < 1:
            raise RuntimeError("MipsEngine needs at least one GPU; there is no CPU search path")
        self.dim = int(dim)
        self.storage_dtype = storage_dtype
        self.devices = [int(d) for d in devices]
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        check(lib.artmips_create(arr, len(self.devices), self.dim, _abi_dtype(storage_dtype),
                                 ctypes.byref(self._h)))
        self.n_rows = 0
        # close() must not destroy the handle under a download chunk running on another thread
        # (index.save_flat(background=True)); later chunks of that export find the engine closed
        self._dl_lock = threading.Lock()

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        lock = getattr(self, "_dl_lock", None)
        if lock is None:  # constructor failed before the handle existed
            return
        with lock:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib.artmips_destroy(self._h)
                self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- building ---------------------------------------------------------------------------
    def begin(self, n_rows):
        check(lib.artmips_begin(self._h, int(n_rows)))
        self._building_rows = int(n_rows)

    def upload_host(self, rows, ids):
        """Append host rows (numpy float16 array, or a CPU torch tensor in the storage dtype)
        and their int32 passage ids."""
        if isinstance(rows, np.ndarray):
            want = {torch.float16: np.float16, torch.float32: np.float32}.get(self.storage_dtype)
            if want is None or rows.dtype != want:
                raise ValueError("numpy rows must match the storage dtype (float16 or float32); "
                                 "pass a CPU torch tensor for bfloat16")
            rows = np.ascontiguousarray(rows)
            n, d = rows.shape
            rows_ptr = rows.ctypes.data
        else:
            if rows.device.type != "cpu" or rows.dtype != self.storage_dtype:
                raise ValueError("host rows must be a CPU tensor of the storage dtype")
            rows = rows.contiguous()
            n, d = rows.shape
            rows_ptr = rows.data_ptr()
        if d != self.dim:
            raise ValueError("rows have dim %d, index has dim %d" % (d, self.dim))
        ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
        if ids.shape != (n,):
            raise ValueError("ids must have shape (%d,)" % n)
        check(lib.artmips_upload(self._h, ctypes.c_void_p(rows_ptr), ctypes.c_void_p(ids.ctypes.data),
                                 n, 0))

    def upload_device(self, slot, rows, ids, row_offset):
        """Copy rows already on a GPU into shard `slot` at row_offset (offset inside the shard)."""
        if rows.dtype != self.storage_dtype or not rows.is_cuda:
            raise ValueError("device rows must be a CUDA tensor of the storage dtype")
        if ids.dtype != torch.int32 or not ids.is_cuda:
            raise ValueError("device ids must be a CUDA int32 tensor")
        rows = rows.contiguous()
        ids = ids.contiguous()
        n, d = rows.shape
        if d != self.dim or ids.shape != (n,):
            raise ValueError("shape mismatch")
        with _on_device(rows.device):
            check(lib.artmips_upload_device(self._h, int(slot), ctypes.c_void_p(rows.data_ptr()),
                                            ctypes.c_void_p(ids.data_ptr()), n, int(row_offset),
                                            _stream_ptr(rows.device)))

    def upload_scatter(self, rows, ids, id_base=1):
        """Scatter a batch of rows on a GPU (float16/bfloat16/float32, converted to the storage
        dtype) into the generation being built: the row with passage id `id` lands at global position
        id - id_base.  Asynchronous on torch's current stream.  The sources may be freed right away:
        torch's allocator reuses memory in the order of that stream, and the library makes that stream
        wait for the staging copies that carry the batch to shards on other devices (they run on the
        shards' own streams) before anything queued later on it can run."""
        if not rows.is_cuda or not ids.is_cuda or rows.device != ids.device:
            raise ValueError("rows and ids must live on the same GPU")
        if ids.dtype != torch.int64:
            ids = ids.to(torch.int64)
        rows = rows.contiguous()
        ids = ids.contiguous()
        n, d = rows.shape
        if d != self.dim or ids.shape != (n,):
            raise ValueError("shape mismatch")
        with _on_device(rows.device):
            check(lib.artmips_upload_scatter(self._h, rows.device.index, ctypes.c_void_p(rows.data_ptr()),
                                             _abi_dtype(rows.dtype), ctypes.c_void_p(ids.data_ptr()), n,
                                             int(id_base), _stream_ptr(rows.device)))

    def finalize(self):
        check(lib.artmips_finalize(self._h))
        self.n_rows = self._building_rows

    def build(self, rows, ids):
        """begin + upload_host + finalize in one call."""
        self.begin(rows.shape[0])
        self.upload_host(rows, ids)
        self.finalize()

    # ---- searching --------------------------------------------------------------------------
    def _check_queries(self, queries, device_index):
        if not isinstance(queries, torch.Tensor):
            raise ValueError("query_embeds must be a torch tensor")
        if queries.dim() != 2 or queries.shape[1] != self.dim:
            raise ValueError("query_embeds must have shape [Q, %d], got %s" % (self.dim, tuple(queries.shape)))
        if queries.shape[0] < 1:
            raise ValueError("query_embeds is empty")
        if not queries.is_cuda:
            raise ValueError("query_embeds must live on a GPU")
        if queries.device.index != device_index:
            raise ValueError("query_embeds is on cuda:%d but the index expects cuda:%d"
                             % (queries.device.index, device_index))
        _abi_dtype(queries.dtype)
        return queries.contiguous()

    def search(self, queries, top_k, score_dtype=torch.float16):
        """(scores [Q,k] score_dtype, ids [Q,k] int32), rows sorted by (score desc, id asc)."""
        top_k = int(top_k)
        if top_k < 1 or top_k > self.n_rows:
            raise ValueError("top_k must be in [1, %d], got %d" % (self.n_rows, top_k))
        q = self._check_queries(queries, self.devices[0])
        dev = q.device
        scores = torch.empty((q.shape[0], top_k), dtype=score_dtype, device=dev)
        ids = torch.empty((q.shape[0], top_k), dtype=torch.int32, device=dev)
        with _on_device(dev):
            check(lib.artmips_search(self._h, ctypes.c_void_p(q.data_ptr()), _abi_dtype(q.dtype),
                                     q.shape[0], top_k, ctypes.c_void_p(scores.data_ptr()),
                                     _abi_dtype(score_dtype), ctypes.c_void_p(ids.data_ptr()),
                                     _stream_ptr(dev)))
        return scores, ids

    def search_async(self, queries, top_k, score_dtype=torch.float16, out_status=None):
        """search() without the host wait and without the overflow retry (artmips_search_async):
        returns (scores, ids, status) with status a CUDA int32 [4] tensor whose word 0 is valid once
        the current stream has passed the call (bit 0: a candidate list overflowed -> grow_lists() and
        repeat; bit 1: internal scratch too small)."""
        top_k = int(top_k)
        if top_k < 1 or top_k > self.n_rows:
            raise ValueError("top_k must be in [1, %d], got %d" % (self.n_rows, top_k))
        q = self._check_queries(queries, self.devices[0])
        dev = q.device
        scores = torch.empty((q.shape[0], top_k), dtype=score_dtype, device=dev)
        ids = torch.empty((q.shape[0], top_k), dtype=torch.int32, device=dev)
        if out_status is None:
            out_status = torch.empty((4,), dtype=torch.int32, device=dev)
        elif out_status.dtype != torch.int32 or out_status.numel() < 4 or out_status.device != dev:
            raise ValueError("out_status must be a CUDA int32 tensor of 4 words on the query's device")
        with _on_device(dev):
            check(lib.artmips_search_async(self._h, ctypes.c_void_p(q.data_ptr()), _abi_dtype(q.dtype),
                                           q.shape[0], top_k, ctypes.c_void_p(scores.data_ptr()),
                                           _abi_dtype(score_dtype), ctypes.c_void_p(ids.data_ptr()),
                                           ctypes.c_void_p(out_status.data_ptr()), _stream_ptr(dev)))
        if q is not queries:
            q.record_stream(torch.cuda.current_stream(dev))
        return scores, ids, out_status

    def search_local(self, slot, queries, top_k):
        """Top-k of one shard: (scores fp32 [Q,k], ids int32 [Q,k]); short shards pad with
        (-inf, INT32_MAX)."""
        top_k = int(top_k)
        if top_k < 1:
            raise ValueError("top_k must be >= 1")
        q = self._check_queries(queries, self.devices[slot])
        dev = q.device
        scores = torch.empty((q.shape[0], top_k), dtype=torch.float32, device=dev)
        ids = torch.empty((q.shape[0], top_k), dtype=torch.int32, device=dev)
        with _on_device(dev):
            check(lib.artmips_search_local(self._h, int(slot), ctypes.c_void_p(q.data_ptr()),
                                           _abi_dtype(q.dtype), q.shape[0], top_k,
                                           ctypes.c_void_p(scores.data_ptr()),
                                           ctypes.c_void_p(ids.data_ptr()), _stream_ptr(dev)))
        return scores, ids

    @staticmethod
    def merge(scores, ids, score_dtype=torch.float16):
        """k-way merge of [G, Q, k] candidate lists (fp32 scores, int32 ids) on their device."""
        if scores.dim() != 3 or scores.shape != ids.shape:
            raise ValueError("scores and ids must both be [G, Q, k]")
        if scores.dtype != torch.float32 or ids.dtype != torch.int32:
            raise ValueError("merge takes float32 scores and int32 ids")
        if not scores.is_cuda or scores.device != ids.device:
            raise ValueError("merge inputs must live on one GPU")
        scores = scores.contiguous()
        ids = ids.contiguous()
        g, nq, k = scores.shape
        dev = scores.device
        out_s = torch.empty((nq, k), dtype=score_dtype, device=dev)
        out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
        with _on_device(dev):
            check(lib.artmips_merge(dev.index, ctypes.c_void_p(scores.data_ptr()),
                                    ctypes.c_void_p(ids.data_ptr()), g, nq, k,
                                    ctypes.c_void_p(out_s.data_ptr()), _abi_dtype(score_dtype),
                                    ctypes.c_void_p(out_i.data_ptr()), _stream_ptr(dev)))
        return out_s, out_i

    # ---- SPMD step without a host round trip (include/artmips.h, "the SPMD step ...") ----------
    @staticmethod
    def packed_words(n_q, top_k):
        return 2 * int(n_q) * int(top_k) + _lib.PACK_TAIL  # = artmips_packed_words(n_q, top_k)

    def search_local_async(self, slot, queries, top_k, out_packed):
        """Enqueue the local top-k of shard `slot` into out_packed (int32 [packed_words(Q, k)]:
        score bits | ids | status tail) on torch's current stream.  Does not wait, does not retry."""
        top_k = int(top_k)
        if top_k < 1:
            raise ValueError("top_k must be >= 1")
        q = self._check_queries(queries, self.devices[slot])
        if (out_packed.dtype != torch.int32 or not out_packed.is_cuda or not out_packed.is_contiguous()
                or out_packed.device != q.device
                or out_packed.numel() != self.packed_words(q.shape[0], top_k)):
            raise ValueError("out_packed must be a contiguous CUDA int32 tensor of packed_words(Q, k) words")
        with _on_device(q.device):
            check(lib.artmips_search_local_async(self._h, int(slot), ctypes.c_void_p(q.data_ptr()),
                                                 _abi_dtype(q.dtype), q.shape[0], top_k,
                                                 ctypes.c_void_p(out_packed.data_ptr()),
                                                 _stream_ptr(q.device)))
        if q is not queries:  # a contiguous copy the library reads after this returns: keep it alive
            q.record_stream(torch.cuda.current_stream(q.device))

    def search_local_noalloc(self, slot, queries, top_k, out_packed):
        """search_local_async for callers that cannot allocate (artmips_search_local_noalloc): the
        shard's workspace is a static 64 MiB device array, shapes that do not fit are refused, calls
        that overlap on one GPU race.  `queries` must be contiguous already (nothing is copied)."""
        top_k = int(top_k)
        if top_k < 1:
            raise ValueError("top_k must be >= 1")
        if not queries.is_contiguous():
            raise ValueError("queries must be contiguous: this entry point copies nothing")
        q = self._check_queries(queries, self.devices[slot])
        if (out_packed.dtype != torch.int32 or not out_packed.is_cuda or not out_packed.is_contiguous()
                or out_packed.device != q.device
                or out_packed.numel() != self.packed_words(q.shape[0], top_k)):
            raise ValueError("out_packed must be a contiguous CUDA int32 tensor of packed_words(Q, k) words")
        with _on_device(q.device):
            check(lib.artmips_search_local_noalloc(self._h, int(slot), ctypes.c_void_p(q.data_ptr()),
                                                   _abi_dtype(q.dtype), q.shape[0], top_k,
                                                   ctypes.c_void_p(out_packed.data_ptr()),
                                                   _stream_ptr(q.device)))

    @staticmethod
    def merge_packed(packed, n_q, top_k, score_dtype=torch.float16, out_flag=None):
        """Merge [G, packed_words(Q, k)] packed lists; out_flag (CUDA int32 [1]) receives the OR of
        their status words.  Asynchronous."""
        if packed.dim() != 2 or packed.dtype != torch.int32 or not packed.is_cuda or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous CUDA int32 tensor [G, words]")
        if packed.shape[1] != MipsEngine.packed_words(n_q, top_k):
            raise ValueError("packed lists have %d words, expected %d"
                             % (packed.shape[1], MipsEngine.packed_words(n_q, top_k)))
        dev = packed.device
        out_s = torch.empty((n_q, top_k), dtype=score_dtype, device=dev)
        out_i = torch.empty((n_q, top_k), dtype=torch.int32, device=dev)
        flag_ptr = ctypes.c_void_p(out_flag.data_ptr()) if out_flag is not None else ctypes.c_void_p()
        with _on_device(dev):
            check(lib.artmips_merge_packed(dev.index, ctypes.c_void_p(packed.data_ptr()), packed.shape[0],
                                           int(n_q), int(top_k), ctypes.c_void_p(out_s.data_ptr()),
                                           _abi_dtype(score_dtype), ctypes.c_void_p(out_i.data_ptr()),
                                           flag_ptr, _stream_ptr(dev)))
        return out_s, out_i

    @staticmethod
    def merge_packed_sync(packed, n_q, top_k, score_dtype=torch.float16):
        """merge_packed plus the step's one host wait: returns (scores, ids, flag) with flag the OR of
        the lists' status words, read from a pinned host word the merge kernel wrote (no device-to-host
        copy is queued).  The current stream is synchronised on return."""
        if packed.dim() != 2 or packed.dtype != torch.int32 or not packed.is_cuda or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous CUDA int32 tensor [G, words]")
        if packed.shape[1] != MipsEngine.packed_words(n_q, top_k):
            raise ValueError("packed lists have %d words, expected %d"
                             % (packed.shape[1], MipsEngine.packed_words(n_q, top_k)))
        dev = packed.device
        out_s = torch.empty((n_q, top_k), dtype=score_dtype, device=dev)
        out_i = torch.empty((n_q, top_k), dtype=torch.int32, device=dev)
        flag = ctypes.c_int32(0)
        with _on_device(dev):
            check(lib.artmips_merge_packed_sync(dev.index, ctypes.c_void_p(packed.data_ptr()), packed.shape[0],
                                                int(n_q), int(top_k), ctypes.c_void_p(out_s.data_ptr()),
                                                _abi_dtype(score_dtype), ctypes.c_void_p(out_i.data_ptr()),
                                                ctypes.byref(flag), _stream_ptr(dev)))
        return out_s, out_i, flag.value

    def fetch_rows(self, positions):
        """Rows of the live index at the given global row positions, as a float32 numpy array
        [n, dim] (faiss's reconstruct, art_index.py:190-193)."""
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.int64).reshape(-1))
        out = np.empty((pos.shape[0], self.dim), dtype=np.float32)
        check(lib.artmips_fetch_rows(self._h, ctypes.c_void_p(pos.ctypes.data), pos.shape[0],
                                     ctypes.c_void_p(out.ctypes.data)))
        return out

    def generation(self):
        """Counter that increases with every finalize(); pair with download()."""
        out = ctypes.c_uint64(0)
        with self._dl_lock:
            if not self._h.value:
                raise ArtmipsError(-3, "the engine is closed")
            check(lib.artmips_generation(self._h, ctypes.byref(out)))
        return out.value

    def download(self, row_offset, n_rows, generation=None):
        """Rows [row_offset, row_offset + n_rows) of the live index and their passage ids, copied to
        the host: (rows numpy [n, dim] in the storage dtype -- bfloat16 as raw uint16 -- , ids int32
        [n]).  May be called from a second thread beside searches; fails if the index is replaced
        (finalize) after `generation` was read."""
        if generation is None:
            generation = self.generation()
        np_dt = {torch.float16: np.float16, torch.bfloat16: np.uint16, torch.float32: np.float32}[self.storage_dtype]
        rows = np.empty((int(n_rows), self.dim), dtype=np_dt)
        ids = np.empty((int(n_rows),), dtype=np.int32)
        with self._dl_lock:
            if not self._h.value:
                raise ArtmipsError(-3, "the engine was closed while its index was being downloaded")
            check(lib.artmips_download(self._h, ctypes.c_uint64(generation), int(row_offset), int(n_rows),
                                       ctypes.c_void_p(rows.ctypes.data), ctypes.c_void_p(ids.ctypes.data)))
        return rows, ids

    def grow_lists(self):
        check(lib.artmips_grow_lists(self._h))

    def set_stats(self, enabled):
        """HIP-event timing of the pipeline on/off (info() keeps the last timed call when off)."""
        check(lib.artmips_set_stats(self._h, 1 if enabled else 0))

    def info(self):
        out = _lib.ArtmipsInfo()
        check(lib.artmips_info(self._h, ctypes.byref(out)))
        return {
            "n_dev": out.n_dev,
            "dim": out.dim,
            "storage_dtype": out.storage_dtype,
            "n_cu": out.n_cu,
            "n_rows": out.n_rows,
            "rows_per_shard": [out.rows_per_shard[i] for i in range(out.n_dev)],
            "index_bytes": out.index_bytes,
            "workspace_bytes": out.workspace_bytes,
            "last_scan_ms": out.last_scan_ms,
            "last_total_ms": out.last_total_ms,
            "last_scan_launches": out.last_scan_launches,
            "last_retries": out.last_retries,
            "last_candidates": out.last_candidates,
            "last_host_enqueue_us": out.last_host_enqueue_us,
            "last_host_sync_us": out.last_host_sync_us,
        }


__all__ = ["MipsEngine", "ArtmipsError"]
