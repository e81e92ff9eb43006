"""Evidence token datasets and the batched id -> tokens fetch that follows every search
(SURVEY.md §8f row 4).

The reference keeps four mmap token datasets (BERT and T0 tokenisations of passage text and
title, ``megatron/model/art_model.py:305-319``) in the ``MMapIndexedDataset`` format of
``megatron/data/indexed_dataset.py:335-512`` and, after each search, reads ``Q*k*4`` items one by
one in a Python loop (``art_model.py:367-381``).  Here:

  * ``MMapTokenDataset`` reads the same ``.idx``/``.bin`` pair (byte-compatible, pinned by
    tests/golden/fx_tokens which the reference's own builder wrote) and adds ``get_batch``;
  * ``DeviceTokenStore`` keeps a dataset's tokens in HBM (Wikipedia: 21M passages x ~100 uint16
    tokens = 4 GB per dataset, nothing next to 288 GB) and gathers the rows of all winners into a
    padded ``[n, max_len]`` int32 tensor with ONE kernel launch (libarttok.so, include/arttok.h);
  * ``fetch_topk_data`` rebuilds the exact nested-list structure ``get_topk`` returns, from four
    device gathers and four host copies instead of ``Q*k*4`` mmap reads.
"""
import ctypes
import os
import struct

import numpy as np
import torch

_HDR_MAGIC = b"MMIDIDX\x00\x00"  # indexed_dataset.py:337
# dtype codes of indexed_dataset.py:92-101 (6 is numpy's removed `np.float` alias = float64)
_DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float64,
           8: np.uint16}
_CODES = {np.dtype(np.uint8): 1, np.dtype(np.int8): 2, np.dtype(np.int16): 3, np.dtype(np.int32): 4,
          np.dtype(np.int64): 5, np.dtype(np.float64): 7, np.dtype(np.uint16): 8}


def index_file_path(prefix_path):
    return prefix_path + ".idx"


def data_file_path(prefix_path):
    return prefix_path + ".bin"


def write_mmap_dataset(prefix_path, items, dtype=np.uint16):
    """Write items (sequences of ints) as an MMapIndexedDataset, one document per item: the bytes
    MMapIndexedDatasetBuilder.add_item/end_document/finalize produce (indexed_dataset.py:341-383,
    539-575)."""
    dtype = np.dtype(dtype)
    sizes = []
    with open(data_file_path(prefix_path), "wb") as f:
        for it in items:
            arr = np.asarray(it, dtype=dtype)
            f.write(arr.tobytes(order="C"))
            sizes.append(arr.size)
    sizes = np.asarray(sizes, dtype=np.int32)
    pointers = np.zeros(len(sizes), dtype=np.int64)
    if len(sizes) > 1:
        np.cumsum(sizes[:-1].astype(np.int64) * dtype.itemsize, out=pointers[1:])
    doc_idx = np.arange(len(sizes) + 1, dtype=np.int64)
    with open(index_file_path(prefix_path), "wb") as f:
        f.write(_HDR_MAGIC)
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<B", _CODES[dtype]))
        f.write(struct.pack("<Q", len(sizes)))
        f.write(struct.pack("<Q", len(doc_idx)))
        f.write(sizes.tobytes(order="C"))
        f.write(pointers.tobytes(order="C"))
        f.write(doc_idx.tobytes(order="C"))


class MMapTokenDataset(object):
    """Read-only view of an MMapIndexedDataset (indexed_dataset.py:385-512)."""

    def __init__(self, path, skip_warmup=True):
        self._path = path
        with open(index_file_path(path), "rb") as stream:
            magic = stream.read(9)
            if magic != _HDR_MAGIC:
                raise ValueError("Index file doesn't match expected format. "
                                 "Make sure that --dataset-impl is configured properly.")
            version, = struct.unpack("<Q", stream.read(8))
            if version != 1:
                raise ValueError("unsupported index version %d" % version)
            dtype_code, = struct.unpack("<B", stream.read(1))
            self._dtype = np.dtype(_DTYPES[dtype_code])
            self._len, = struct.unpack("<Q", stream.read(8))
            self._doc_count, = struct.unpack("<Q", stream.read(8))
            offset = stream.tell()
        self._index_mmap = np.memmap(index_file_path(path), mode="r", order="C")
        self._sizes = np.frombuffer(self._index_mmap, dtype=np.int32, count=self._len, offset=offset)
        self._pointers = np.frombuffer(self._index_mmap, dtype=np.int64, count=self._len,
                                       offset=offset + self._sizes.nbytes)
        self._doc_idx = np.frombuffer(self._index_mmap, dtype=np.int64, count=self._doc_count,
                                      offset=offset + self._sizes.nbytes + self._pointers.nbytes)
        if os.path.getsize(data_file_path(path)) > 0:
            self._bin = np.memmap(data_file_path(path), mode="r", order="C")
        else:
            self._bin = np.zeros(0, dtype=np.uint8)
        if not skip_warmup:
            self._bin.sum()  # touch every page once (indexed_dataset.py:_warmup_mmap_file)

    dtype = property(lambda self: self._dtype)
    sizes = property(lambda self: self._sizes)
    pointers = property(lambda self: self._pointers)
    doc_idx = property(lambda self: self._doc_idx)

    def __len__(self):
        return self._len

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            if idx < 0:
                idx += self._len
            ptr, size = int(self._pointers[idx]), int(self._sizes[idx])
            return np.frombuffer(self._bin, dtype=self._dtype, count=size, offset=ptr)
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self._len)
            if step != 1:
                raise ValueError("Slices into indexed_dataset must be contiguous")
            return [self[i] for i in range(start, stop)]
        raise TypeError("index must be an int or a contiguous slice")

    def get(self, idx, offset=0, length=None):
        ptr, size = int(self._pointers[idx]), int(self._sizes[idx])
        if length is None:
            length = size - offset
        return np.frombuffer(self._bin, dtype=self._dtype, count=length,
                             offset=ptr + offset * self._dtype.itemsize)

    def tokens(self):
        """The whole .bin as one array of the dataset's dtype (a view of the mmap)."""
        return np.frombuffer(self._bin, dtype=self._dtype, count=self._bin.nbytes // self._dtype.itemsize)

    def get_batch(self, indices):
        """Items `indices` (any order, repeats allowed) as (flat tokens, offsets [n+1]): item j of
        the batch is flat[offsets[j]:offsets[j+1]].  One vectorised gather, no per-item Python."""
        indices = np.asarray(indices, dtype=np.int64).ravel()
        if indices.size and (indices.min() < 0 or indices.max() >= self._len):
            raise IndexError("item index out of range")
        sizes = self._sizes[indices].astype(np.int64)
        offsets = np.zeros(indices.size + 1, dtype=np.int64)
        np.cumsum(sizes, out=offsets[1:])
        starts = self._pointers[indices] // self._dtype.itemsize
        # position p of the flat output belongs to batch item j = searchsorted(offsets, p) - 1
        within = np.arange(offsets[-1], dtype=np.int64) - np.repeat(offsets[:-1], sizes)
        flat = self.tokens()[np.repeat(starts, sizes) + within]
        return flat, offsets


# ---- libarttok.so ---------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
TOK_LIB_PATH = os.path.join(_HERE, "lib", "libarttok.so")
TOK_SYMBOLS = ("arttok_create", "arttok_destroy", "arttok_gather", "arttok_gather_multi", "arttok_info",
               "arttok_last_error")
MAX_SETS = 4  # ARTTOK_MAX_SETS
_toklib = None


def _tok():
    """Load libarttok.so on first use; there is no fallback (fails loudly if it was not built)."""
    global _toklib
    if _toklib is None:
        if not os.path.exists(TOK_LIB_PATH):
            raise ImportError("%s is missing: run art_amd/csrc/build.sh (or __graft_entry__.build())"
                              % TOK_LIB_PATH)
        lib = ctypes.CDLL(TOK_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        lib.arttok_create.argtypes = [i32, vp, i64, i32, vp, vp, i64, ctypes.POINTER(vp)]
        lib.arttok_create.restype = i32
        lib.arttok_destroy.argtypes = [vp]
        lib.arttok_destroy.restype = None
        lib.arttok_gather.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp]
        lib.arttok_gather.restype = i32
        lib.arttok_gather_multi.argtypes = [vp, i32, vp, i64, i32, vp, vp, vp, vp, vp]
        lib.arttok_gather_multi.restype = i32
        lib.arttok_info.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.arttok_info.restype = i32
        lib.arttok_last_error.argtypes = []
        lib.arttok_last_error.restype = ctypes.c_char_p
        _toklib = lib
    return _toklib


class ArttokError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("arttok error %d: %s" % (code, message))
        self.code = code


def _tcheck(rc):
    if rc != 0:
        raise ArttokError(rc, _tok().arttok_last_error().decode("utf-8", "replace"))


class DeviceTokenStore(object):
    """One token dataset resident on a GPU; `gather` fetches padded rows for a batch of ids."""

    def __init__(self, dataset, device=0, id_base=1):
        if dataset.dtype.itemsize not in (1, 2, 4, 8) or dataset.dtype.kind not in "iu":
            raise ValueError("token dataset must hold integers, got %s" % dataset.dtype)
        if dataset.dtype.itemsize <= 2 and dataset.dtype.kind != "u":
            raise ValueError("1- and 2-byte token datasets must be unsigned, got %s" % dataset.dtype)
        self.device = int(device)
        self.id_base = int(id_base)  # art_model.py:374 looks items up at idx - 1
        self.n_items = len(dataset)
        self.max_item_len = int(dataset.sizes.max()) if self.n_items else 0
        toks = np.ascontiguousarray(dataset.tokens())
        ptrs = np.ascontiguousarray(dataset.pointers, dtype=np.int64)
        sizes = np.ascontiguousarray(dataset.sizes, dtype=np.int32)
        self._h = ctypes.c_void_p()
        _tcheck(_tok().arttok_create(self.device, ctypes.c_void_p(toks.ctypes.data if toks.size else 0),
                                     toks.size, dataset.dtype.itemsize, ctypes.c_void_p(ptrs.ctypes.data),
                                     ctypes.c_void_p(sizes.ctypes.data), self.n_items, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _tok().arttok_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather(self, ids, max_len=None, pad=0):
        """ids: CUDA int32 tensor of passage ids (any shape, e.g. topkindex [Q, k]).
        Returns (tokens [..., max_len] int32, lengths [...] int32) on the store's device; rows are
        truncated to max_len (default: the longest item of the dataset), an id outside the dataset
        gives length -1.  Asynchronous on torch's current stream."""
        if not ids.is_cuda or ids.dtype != torch.int32 or ids.device.index != self.device:
            raise ValueError("ids must be a CUDA int32 tensor on cuda:%d" % self.device)
        if ids.numel() < 1:
            raise ValueError("ids is empty")
        max_len = int(max_len) if max_len is not None else max(1, self.max_item_len)
        flat = ids.contiguous().view(-1)
        tokens = torch.empty((flat.numel(), max_len), dtype=torch.int32, device=ids.device)
        lens = torch.empty((flat.numel(),), dtype=torch.int32, device=ids.device)
        with torch.cuda.device(ids.device):
            _tcheck(_tok().arttok_gather(self._h, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                         self.id_base, max_len, int(pad), ctypes.c_void_p(tokens.data_ptr()),
                                         ctypes.c_void_p(lens.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream)))
        return tokens.view(tuple(ids.shape) + (max_len,)), lens.view(tuple(ids.shape))

    def info(self):
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _tcheck(_tok().arttok_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"n_items": a.value, "n_tokens": b.value, "device_bytes": c.value}


def gather_multi(stores, ids, max_lens=None, pads=None):
    """`store.gather(ids)` for up to four DeviceTokenStore objects on one device in ONE launch
    (arttok_gather_multi): the four token datasets a training step reads for the same [Q, k] winners
    (art_model.py:367-381).  Returns a list of (tokens [..., max_len] int32, lengths [...] int32), one
    per store.  Asynchronous on torch's current stream."""
    stores = list(stores)
    if not 1 <= len(stores) <= MAX_SETS:
        raise ValueError("gather_multi takes 1 to %d stores" % MAX_SETS)
    dev, id_base = stores[0].device, stores[0].id_base
    if any(st.device != dev or st.id_base != id_base for st in stores):
        raise ValueError("the stores of one gather_multi call must share a device and an id base")
    if not ids.is_cuda or ids.dtype != torch.int32 or ids.device.index != dev:
        raise ValueError("ids must be a CUDA int32 tensor on cuda:%d" % dev)
    if ids.numel() < 1:
        raise ValueError("ids is empty")
    n_sets = len(stores)
    if max_lens is None:
        max_lens = [None] * n_sets
    max_lens = [int(m) if m is not None else max(1, st.max_item_len) for m, st in zip(max_lens, stores)]
    pads = [0] * n_sets if pads is None else [int(x) for x in pads]
    flat = ids.contiguous().view(-1)
    toks = [torch.empty((flat.numel(), m), dtype=torch.int32, device=ids.device) for m in max_lens]
    lens = [torch.empty((flat.numel(),), dtype=torch.int32, device=ids.device) for _ in stores]
    vp = ctypes.c_void_p
    handles = (vp * n_sets)(*[st._h.value for st in stores])
    out_t = (vp * n_sets)(*[t.data_ptr() for t in toks])
    out_l = (vp * n_sets)(*[t.data_ptr() for t in lens])
    ml = (ctypes.c_int32 * n_sets)(*max_lens)
    pd = (ctypes.c_int32 * n_sets)(*pads)
    with torch.cuda.device(ids.device):
        _tcheck(_tok().arttok_gather_multi(handles, n_sets, vp(flat.data_ptr()), flat.numel(), id_base, ml, pd,
                                           out_t, out_l, vp(torch.cuda.current_stream(ids.device).cuda_stream)))
    return [(t.view(tuple(ids.shape) + (m,)), l.view(tuple(ids.shape))) for t, l, m in zip(toks, lens, max_lens)]


def _rows_to_lists(tokens, lens):
    """[n, L] int32 + [n] lengths (host numpy) -> list of n Python lists (one bulk tolist)."""
    if (lens < 0).any():
        raise IndexError("a passage id lies outside the token dataset")
    flat = tokens[np.arange(tokens.shape[1])[None, :] < lens[:, None]].tolist()
    out, pos = [], 0
    for n in lens.tolist():
        out.append(flat[pos:pos + n])
        pos += n
    return out


def fetch_topk_data(topkindex, passages_bert, title_bert, passages_t0, title_t0):
    """The token lookup of ``get_topk`` (art_model.py:367-381) for a [Q, k] int32 CUDA id tensor.

    The four arguments are DeviceTokenStore objects.  Returns the reference's structure: a list with,
    per query, ``(topkarray, [(doctext_bert, title_bert), ...], [(doctext_t0, title_t0), ...])`` where
    every leaf is a Python list of ints."""
    q, k = topkindex.shape
    gathered = gather_multi((passages_bert, title_bert, passages_t0, title_t0), topkindex)
    ids_host = topkindex.cpu().numpy()
    lists = [_rows_to_lists(t.view(q * k, -1).cpu().numpy(), l.view(-1).cpu().numpy()) for t, l in gathered]
    pb, tb, p0, t0 = lists
    topk_data = []
    for r in range(q):
        lo = r * k
        topk_data.append((ids_host[r].tolist(),
                          [(pb[lo + j], tb[lo + j]) for j in range(k)],
                          [(p0[lo + j], t0[lo + j]) for j in range(k)]))
    return topk_data


def fetch_topk_data_host(topkindex, passages_bert, title_bert, passages_t0, title_t0, id_base=1):
    """Same structure from MMapTokenDataset objects on the host (``get_batch`` per dataset): for
    set-ups that leave the tokens on disk.  topkindex: [Q, k] array-like of passage ids."""
    ids = np.asarray(topkindex.cpu() if isinstance(topkindex, torch.Tensor) else topkindex, dtype=np.int64)
    q, k = ids.shape
    lists = []
    for ds in (passages_bert, title_bert, passages_t0, title_t0):
        flat, offs = ds.get_batch(ids.ravel() - id_base)
        flat = flat.tolist()
        offs = offs.tolist()
        lists.append([flat[offs[j]:offs[j + 1]] for j in range(q * k)])
    pb, tb, p0, t0 = lists
    return [(ids[r].tolist(),
             [(pb[r * k + j], tb[r * k + j]) for j in range(k)],
             [(p0[r * k + j], t0[r * k + j]) for j in range(k)]) for r in range(q)]


__all__ = ["MMapTokenDataset", "write_mmap_dataset", "DeviceTokenStore", "gather_multi", "fetch_topk_data",
           "fetch_topk_data_host", "ArttokError"]
