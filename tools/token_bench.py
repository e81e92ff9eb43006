#!/usr/bin/env python
"""id -> token fetch after a search (art_model.py:367-381): the reference's per-item loop over mmap
datasets against the HBM-resident store, on synthetic datasets (N items, ragged uint16/int32 rows),
for the training shape Q=64, k=32.  Prints one JSON line."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd import token_store as ts  # noqa: E402

n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
rng = np.random.Generator(np.random.PCG64(3))
tmp = tempfile.mkdtemp()
specs = {"passages_bert": (np.uint16, 30522, 60, 140), "title_bert": (np.uint16, 30522, 2, 12),
         "passages_t0": (np.int32, 32100, 70, 160), "title_t0": (np.int32, 32100, 2, 16)}
data = {}
for name, (dt, vocab, lo, hi) in specs.items():
    sizes = rng.integers(lo, hi, size=n_items)
    flat = rng.integers(0, vocab, size=int(sizes.sum())).astype(dt)
    prefix = os.path.join(tmp, name)
    # same bytes as write_mmap_dataset, written in bulk
    flat.tofile(prefix + ".bin")
    ts.write_mmap_dataset(prefix + "_hdr", [[]], dtype=dt)  # header layout check only
    import struct
    ptr = np.zeros(n_items, dtype=np.int64)
    np.cumsum(sizes[:-1].astype(np.int64) * np.dtype(dt).itemsize, out=ptr[1:])
    with open(prefix + ".idx", "wb") as f:
        f.write(b"MMIDIDX\x00\x00" + struct.pack("<Q", 1) + struct.pack("<B", 8 if dt == np.uint16 else 4))
        f.write(struct.pack("<Q", n_items) + struct.pack("<Q", n_items + 1))
        f.write(sizes.astype(np.int32).tobytes() + ptr.tobytes() + np.arange(n_items + 1, dtype=np.int64).tobytes())
    data[name] = ts.MMapTokenDataset(prefix)
names = list(specs)
ids = rng.integers(1, n_items + 1, size=(64, 32)).astype(np.int32)


def reference_loop():
    """art_model.py:367-381 restated over the same reader"""
    out = []
    for topkarray in ids:
        topkarray = topkarray.tolist()
        a, b = [], []
        for idx in topkarray:
            a.append((data["passages_bert"][idx - 1].tolist(), data["title_bert"][idx - 1].tolist()))
            b.append((data["passages_t0"][idx - 1].tolist(), data["title_t0"][idx - 1].tolist()))
        out.append((topkarray, a, b))
    return out


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r

t_ref, want = timed(reference_loop, 5)
t_host, got_h = timed(lambda: ts.fetch_topk_data_host(ids, *[data[n] for n in names]), 5)
stores = [ts.DeviceTokenStore(data[n], device=0) for n in names]
ids_dev = torch.from_numpy(ids).cuda()
t_dev, got_d = timed(lambda: ts.fetch_topk_data(ids_dev, *stores), 20)


def gather_only():
    outs = [s.gather(ids_dev) for s in stores]
    torch.cuda.synchronize()
    return outs

t_gather, _ = timed(gather_only, 50)


def gather_fused():
    outs = ts.gather_multi(stores, ids_dev)
    torch.cuda.synchronize()
    return outs

t_fused, _ = timed(gather_fused, 50)


def events_ms(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

t_gather_ev = events_ms(lambda: [s.gather(ids_dev) for s in stores])
t_fused_ev = events_ms(lambda: ts.gather_multi(stores, ids_dev))

# evaluation-set shape (3,610 queries x top-100 ids): the gather as HBM-bound byte work, timed by events
big = torch.from_numpy(rng.integers(1, n_items + 1, size=(3610, 100)).astype(np.int32)).cuda()
big_lines = {}
for name, store in zip(names, stores):
    max_len = (store.max_item_len + 3) // 4 * 4    # 16-byte aligned rows: the int4-store kernel
    store.gather(big, max_len=max_len)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        tok, lens = store.gather(big, max_len=max_len)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    elem = 2 if specs[name][0] == np.uint16 else 4
    moved = float(lens.clamp(min=0).sum().item()) * elem + big.numel() * (max_len * 4 + 4 + 4 + 12)
    big_lines[name] = {"ms": ms, "GB_moved": moved / 1e9, "GB_per_s": moved / ms / 1e6, "frac_of_8TBs": moved / ms / 1e6 / 8000.0}
# the four datasets of the evaluation shape in one launch against four launches (device time, events)
big_lens = [(st.max_item_len + 3) // 4 * 4 for st in stores]
four_ms = events_ms(lambda: [st.gather(big, max_len=m) for st, m in zip(stores, big_lens)], 10)
fused_ms = events_ms(lambda: ts.gather_multi(stores, big, max_lens=big_lens), 10)
moved_all = sum(v["GB_moved"] for v in big_lines.values())
norm = lambda x: json.loads(json.dumps(x))
assert norm(got_h) == norm(want) and norm(got_d) == norm(want)
print(json.dumps({"items_per_dataset": n_items, "ids": "64 x 32", "reference_loop_ms": t_ref * 1e3,
                  "host_batched_ms": t_host * 1e3, "device_store_to_lists_ms": t_dev * 1e3,
                  "device_gather_padded_tensors_ms": t_gather * 1e3,
                  "device_gather_fused_one_launch_ms": t_fused * 1e3,
                  "device_time_four_launches_ms": t_gather_ev, "device_time_fused_ms": t_fused_ev,
                  "eval_shape_four_datasets": {"four_launches_ms": four_ms, "fused_ms": fused_ms, "GB_moved": moved_all,
                                               "fused_GB_per_s": moved_all / fused_ms * 1e3,
                                               "fused_frac_of_8TBs": moved_all / fused_ms * 1e3 / 8000.0},
                  "device_bytes": sum(s.info()["device_bytes"] for s in stores),
                  "eval_shape_gather_3610x100": big_lines}))
