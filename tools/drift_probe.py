#!/usr/bin/env python
"""Does the scan time drift while the GPU stays busy?  Prints the FILTER-scan time of every 50th step
over N steps of the headline shape, in one process on one index (and the GPU clock if readable)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.float16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
t0 = time.perf_counter()
acc = []
for i in range(n_steps):
    eng.search(q, 32)
    acc.append(eng.info()["last_scan_ms"])
    if (i + 1) % 50 == 0:
        a = sorted(acc)
        print("t=%6.2f s  steps %4d-%4d: scan median %.3f ms  min %.3f  max %.3f"
              % (time.perf_counter() - t0, i - 48, i + 1, a[len(a) // 2], a[0], a[-1]), flush=True)
        acc = []
