#!/usr/bin/env python
"""Large-batch searches in a loop, for tools/r3/shared_gpu.sh (two of these side by side share one GPU):
prints median and maximum wall time of a search and checks that every repetition returns the same ids."""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows, nq, k, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = MipsEngine(768, torch.bfloat16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.bfloat16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((nq, 768), generator=g, device="cuda").to(torch.bfloat16)
_, ref = eng.search(q, k)
# wait for the start signal so that both processes loop at the same time
while not os.path.exists(sys.argv[5]):
    time.sleep(0.01)
t = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, ids = eng.search(q, k)
    torch.cuda.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
    assert torch.equal(ids, ref)
print("%d searches of %d x top-%d over %d rows: median %.2f ms, max %.2f ms, ids identical every time" % (
    reps, nq, k, rows, statistics.median(t), max(t)))
