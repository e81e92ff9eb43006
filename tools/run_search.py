#!/usr/bin/env python
"""One search shape on a synthetic index (for rocprofv3 counter passes): run_search.py ROWS Q K [dtype]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[4] if len(sys.argv) > 4 else "f16"]
eng = MipsEngine(768, dt, [0])
bench.build_shard(eng, 0, rows, 1234, dt)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((nq, 768), generator=g, device="cuda").to(dt)
for _ in range(3):
    eng.search(q, k)
i = eng.info()
print("rows %d Q %d k %d: scan %.3f ms over %d sweeps, pipeline %.3f ms" %
      (rows, nq, k, i["last_scan_ms"], i["last_scan_launches"], i["last_total_ms"]))
