#!/usr/bin/env python
"""Throughput of the index-refresh upload paths on one GPU:
  host     artmips_upload from a contiguous host array (pinned double-buffered H2D)
  scatter  artmips_upload_scatter of fp16 / fp32 device batches of 128*world rows (the indexer's batch)
  dict     the reference protocol: dict store -> chunks -> artmips_upload (Python bound)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd.art_index import DistributedBruteForceIndex, OpenRetreivalDataStore  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
dim = 768
rows = np.random.default_rng(0).standard_normal((n, dim), dtype=np.float32).astype(np.float16)
ids = np.arange(1, n + 1, dtype=np.int32)
gb = n * dim * 2 / 1e9

eng = MipsEngine(dim, torch.float16, [0])
eng.build(rows[:1000], ids[:1000])  # warm the pinned buffers
t0 = time.perf_counter()
eng.begin(n); eng.upload_host(rows, ids); eng.finalize()
t = time.perf_counter() - t0
print(json.dumps({"path": "host", "rows": n, "GB": gb, "seconds": t, "GB_per_s": gb / t}), flush=True)

for dt, name in ((torch.float16, "scatter fp16 batches"), (torch.float32, "scatter fp32 batches")):
    d_rows = torch.from_numpy(rows).cuda().to(dt)
    d_ids = torch.arange(1, n + 1, device="cuda", dtype=torch.int64)
    perm = torch.randperm(n, device="cuda")
    batch = 1024
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.begin(n)
    for b in range(0, n, batch):
        sel = perm[b:b + batch]
        eng.upload_scatter(d_rows[sel], d_ids[sel], 1)
    eng.finalize()
    t = time.perf_counter() - t0
    print(json.dumps({"path": name, "rows": n, "batch": batch, "GB_stored": gb, "seconds": t,
                      "rows_per_s": n / t, "GB_per_s_stored": gb / t}), flush=True)
    # the kernel itself (HBM-bound byte work): one 262,144-row batch per call, timed by events around the
    # calls; destinations scattered over the whole shard (a random permutation: every 1.5 KiB row opens a
    # new DRAM page) and in one run of consecutive ids (what the indexer's batches are: dataset order)
    big = 262144
    for order, sel in (("random ids", perm[:big]), ("consecutive ids", torch.arange(big, device="cuda") + n // 3)):
        src, src_ids = d_rows[sel].contiguous(), d_ids[sel].contiguous()
        eng.begin(n)
        eng.upload_scatter(src, src_ids, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        times = []
        for _ in range(reps):
            eng.begin(n)   # fresh generation: every row may be written once only
            e0.record()
            eng.upload_scatter(src, src_ids, 1)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = sorted(times)[reps // 2]
        moved = big * dim * (src.element_size() + 2) + big * 12
        print(json.dumps({"path": name + ", one 262,144-row batch (kernel), " + order, "ms": ms,
                          "GB_moved": moved / 1e9, "GB_per_s": moved / ms / 1e6,
                          "frac_of_8TBs": moved / ms / 1e6 / 8000.0, "rows_per_s": big / ms * 1e3}), flush=True)
    del d_rows

# the streaming SPMD writer (art_amd/refresh.py): the indexer's 1,024-row batches through add_block_data,
# exchange rounds every 64 MiB over RCCL (one rank here: the all_to_alls run, the peer is this rank itself)
import torch.distributed as dist  # noqa: E402
from art_amd.refresh import SpmdIndexWriter  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d_rows = torch.from_numpy(rows).cuda()
d_ids = torch.arange(1, n + 1, device="cuda", dtype=torch.int64)
for always in (False, True):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w = SpmdIndexWriter(eng, n, id_base=1, always_exchange=always)
        for b in range(0, n, 1024):
            w.add_block_data(d_ids[b:b + 1024], d_rows[b:b + 1024])
        w.commit()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print(json.dumps({"path": "SpmdIndexWriter stream, 1,024-row fp16 batches, " +
                              ("RCCL all_to_all rounds (world 1)" if always else "no exchange (world 1)"),
                      "rows": n, "rounds": w.rounds, "seconds": t, "rows_per_s": n / t,
                      "peak_buffered_MB": w.peak_buffered_bytes / 1e6}), flush=True)
del d_rows
dist.destroy_process_group()

m = min(n, 200000)
store = OpenRetreivalDataStore("/tmp/_bench_store.pkl", load_from_path=False, rank=0)
t0 = time.perf_counter()
store.add_block_data(np.arange(1, m + 1, dtype=np.int64), rows[:m])
t_add = time.perf_counter() - t0
t0 = time.perf_counter()
DistributedBruteForceIndex(dim, store, devices=[0])
t_idx = time.perf_counter() - t0
print(json.dumps({"path": "dict store (reference protocol)", "rows": m, "add_block_data_s": t_add,
                  "add_embed_data_s": t_idx, "rows_per_s_upload": m / t_idx}), flush=True)
