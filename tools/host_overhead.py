#!/usr/bin/env python
"""Where does a search call spend host time?  (python wall, C enqueue, C stream wait, GPU pipeline)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from art_amd.engine import MipsEngine
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2626916
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.float16)
g = torch.Generator(device="cuda"); g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
for _ in range(5): eng.search(q, 32)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
enq = syn = 0.0
for _ in range(N):
    eng.search(q, 32)
t1 = time.perf_counter()
for _ in range(20):
    eng.search(q, 32); i = eng.info(); enq += i["last_host_enqueue_us"]; syn += i["last_host_sync_us"]
i = eng.info()
print("rows %d: python wall %.1f us/call | C enqueue %.1f us | C sync wait %.1f us | GPU pipeline %.1f us (scan %.1f)"
      % (rows, (t1 - t0) / N * 1e6, enq / 20, syn / 20, i["last_total_ms"] * 1e3, i["last_scan_ms"] * 1e3))
