#!/usr/bin/env python
"""Gather time of arttok_gather against the padded row length at the evaluation shape (3,610 x 100 ids of a
2M-item synthetic uint16 dataset, 60..139 tokens per item): separates row alignment from bytes moved."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd import token_store as ts  # noqa: E402

n_items = 2000000
rng = np.random.Generator(np.random.PCG64(3))
sizes = rng.integers(60, 140, size=n_items).astype(np.int32)
flat = rng.integers(0, 30522, size=int(sizes.sum())).astype(np.uint16)
ptr = np.zeros(n_items, dtype=np.int64)
np.cumsum(sizes[:-1].astype(np.int64) * 2, out=ptr[1:])


class _Arrays(object):   # the few members DeviceTokenStore reads of a token dataset
    dtype = np.dtype(np.uint16)
    pointers = ptr

    def __len__(self):
        return n_items

    def tokens(self):
        return flat


_Arrays.sizes = sizes
store = ts.DeviceTokenStore(_Arrays(), device=0)
ids = torch.from_numpy(rng.integers(1, n_items + 1, size=(3610, 100)).astype(np.int32)).cuda()
for max_len in (64, 128, 140, 144, 160, 192, 256):
    store.gather(ids, max_len=max_len)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = []
    for _ in range(7):
        e0.record(); store.gather(ids, max_len=max_len); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    ms = sorted(t)[3]
    out_bytes = ids.numel() * max_len * 4
    print(json.dumps({"max_len": max_len, "row_bytes": max_len * 4, "ms": ms, "out_GB_per_s": out_bytes / ms / 1e6}), flush=True)
