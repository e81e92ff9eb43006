#!/usr/bin/env python
"""FILTER-scan and pipeline time against shard size (64 queries, top-32): separates the fixed cost of
a launch from the streaming rate.  size_sweep.py [rows,rows,...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
    [164183, 328366, 656729, 1313458, 2626916, 5253831, 10507662, 21015324]
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
pts = []
for n in sizes:
    eng = MipsEngine(768, torch.float16, [0])
    bench.build_shard(eng, 0, n, 1234, torch.float16)
    scan, tot = [], []
    for i in range(25):
        eng.search(q, 32)
        inf = eng.info()
        if i >= 5:
            scan.append(inf["last_scan_ms"])
            tot.append(inf["last_total_ms"])
    scan.sort()
    tot.sort()
    s, t = scan[len(scan) // 2], tot[len(tot) // 2]
    pts.append((n * 1536.0, s))
    print("rows %9d: scan %8.1f us (%6.0f GB/s)  pipeline %8.1f us  candidates/query %.0f"
          % (n, s * 1e3, n * 1536.0 / s / 1e6, t * 1e3, inf["last_candidates"] / 64), flush=True)
    eng.close()
if len(pts) >= 2:
    (b0, t0), (b1, t1) = pts[0], pts[-1]
    bw = (b1 - b0) / (t1 - t0) / 1e6
    print("two-point fit: %.0f GB/s streaming + %.1f us fixed per launch" % (bw, (t0 - b0 / bw / 1e6) * 1e3))
