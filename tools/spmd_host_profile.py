#!/usr/bin/env python
"""Host time of each call of one SPMD step (one rank, nccl, 1/8 shard): where does the GPU sit idle
between steps?"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29588")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2626916
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.float16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
k = 32
words = MipsEngine.packed_words(64, k)
local = torch.empty(words, dtype=torch.int32, device="cuda")
gathered = torch.empty((1, words), dtype=torch.int32, device="cuda")
names = ["search_local_async", "all_gather_into_tensor", "merge_packed_sync (merge + the step's wait)"]
acc = [0.0] * len(names)
N = 300
eng.set_stats(False)  # as bench.py runs three steps out of four
for it in range(N + 20):
    t = [time.perf_counter()]
    eng.search_local_async(0, q, k, local)
    t.append(time.perf_counter())
    dist.all_gather_into_tensor(gathered, local)
    t.append(time.perf_counter())
    out = MipsEngine.merge_packed_sync(gathered, 64, k, torch.float16)
    t.append(time.perf_counter())
    if it >= 20:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]
eng.set_stats(True)
eng.search_local_async(0, q, k, local)
torch.cuda.synchronize()
tot = sum(acc) / N * 1e6
print("rows %d: step %.1f us on the host" % (rows, tot))
for n, a in zip(names, acc):
    print("  %-24s %7.1f us" % (n, a / N * 1e6))
i = eng.info()
print("GPU pipeline (pack..select) %.1f us, scan %.1f us" % (i["last_total_ms"] * 1e3, i["last_scan_ms"] * 1e3))
dist.destroy_process_group()
