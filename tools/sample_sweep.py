#!/usr/bin/env python
"""Sweep ARTMIPS_SAMPLE_DIV (pass-0 sample = tiles / div) for a few (queries, top_k) shapes on the
full-size index; prints the median HIP-event time of the whole pipeline per setting."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows = int(os.environ.get("ROWS", bench.N_WIKI))
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.float16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1].split(",")]
divs = [int(x) for x in sys.argv[2].split(",")]
for nq, k in shapes:
    q = torch.randn((nq, 768), generator=g, device="cuda").half()
    for d in divs:
        os.environ["ARTMIPS_SAMPLE_DIV"] = str(d)
        tot, scan = [], []
        for r in range(4):
            eng.search(q, k)
            info = eng.info()
            if r:
                tot.append(info["last_total_ms"])
                scan.append(info["last_scan_ms"])
        tot.sort()
        scan.sort()
        print("Q=%d k=%d div=%d: pipeline %.3f ms (scan %.3f)  cand/query %.0f retries %d -> %.0f q/s"
              % (nq, k, d, tot[1], scan[1], info["last_candidates"] / nq, info["last_retries"], nq / tot[1] * 1e3),
              flush=True)
