#!/usr/bin/env python
"""Host cost of the one-process / many-shards search (the reference's process model, art_index.py:204):
tiny shards, so that GPU work is negligible and the wall time of a call is what the host spends
planning, enqueuing and collecting G pipelines.  All shards sit on device 0 (one-GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the ARTMIPS_* knobs this script turns exist only in the diagnostic build of the library
os.environ.setdefault("ARTMIPS_DIAGNOSTICS", "1")
os.environ.setdefault("ARTMIPS_LIBRARY", os.path.join(ROOT, "art_amd", "lib", "libartmips_diag.so"))
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

q = torch.randn((64, 768), device="cuda").half()
for g in (1, 2, 4, 8):
    eng = MipsEngine(768, torch.float16, [0] * g)
    eng.begin(20000 * g)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    for s in range(g):
        rows = (torch.randn((20000, 768), generator=gen, device="cuda") * 0.03).half()
        ids = torch.arange(s * 20000 + 1, (s + 1) * 20000 + 1, device="cuda", dtype=torch.int32)
        eng.upload_device(s, rows, ids, 0)
    eng.finalize()
    eng.set_stats(False)
    for _ in range(20):
        eng.search(q, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.search(q, 32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e6
    eng.set_stats(True)
    eng.search(q, 32)
    print("shards %d: %.1f us per call (20,000 rows per shard); host time until the last shard's pipeline "
          "is enqueued %.1f us" % (g, dt, eng.info()["last_host_enqueue_us"]))
    eng.close()
