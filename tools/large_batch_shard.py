#!/usr/bin/env python
"""One large-batch call on a 1/8 Wikipedia shard (what each GPU of an 8-GPU node runs for BASELINE
config 4): 2,626,916 x 768 bf16 rows, 2,048 queries, top-1000.  Meant to run under rocprofv3
--kernel-trace for the per-kernel breakdown of a sweep."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2626916
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4321)
eng = MipsEngine(768, torch.bfloat16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.bfloat16)
q = torch.randn((2048, 768), generator=g, device=dev).bfloat16()
eng.search(q, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.search(q, k)
torch.cuda.synchronize()
print("rows %d k %d: %.3f ms per 256-query sweep" % (rows, k, (time.perf_counter() - t0) / 3 / 8 * 1e3))
eng.close()
