#!/usr/bin/env python
"""Why does the same scan run 4.5 ms in one process and 4.8 ms in the next?  One measurement per
invocation: build the 21M-row index, time the FILTER scan and a plain device copy.
box_state_probe.py TAG [prealloc_gb]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

tag = sys.argv[1]
pre = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if pre > 0:
    x = torch.empty(int(pre * (1 << 30)), dtype=torch.uint8, device="cuda")
    x.fill_(1)
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
free, total = torch.cuda.mem_get_info()
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.float16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
ts = []
for i in range(25):
    eng.search(q, 32)
    if i >= 5:
        ts.append(eng.info()["last_scan_ms"])
ts.sort()
a = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
b = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    b.copy_(a)
e1.record()
torch.cuda.synchronize()
cp = e0.elapsed_time(e1) / 5
print("%-28s free %.0f GB | scan median %.3f ms min %.3f (%.0f GB/s) | 8 GiB copy %.3f ms (%.0f GB/s r+w)"
      % (tag, free / 1e9, ts[len(ts) // 2], ts[0], bench.N_WIKI * 1536 / ts[len(ts) // 2] / 1e6, cp,
         2 * (8 << 30) / cp / 1e6), flush=True)
