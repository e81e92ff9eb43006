#!/usr/bin/env python
"""Sweep ARTMIPS_SAMPLE_DIV (pass-0 sample = tiles / div) for a few (queries, top_k) shapes on the
full-size index; prints the median HIP-event time of the whole pipeline per setting."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the ARTMIPS_* knobs this script turns exist only in the diagnostic build of the library
os.environ.setdefault("ARTMIPS_DIAGNOSTICS", "1")
os.environ.setdefault("ARTMIPS_LIBRARY", os.path.join(ROOT, "art_amd", "lib", "libartmips_diag.so"))
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
    variants = ["0"]
    for d in divs:
        os.environ["ARTMIPS_SAMPLE_DIV"] = str(d)
        res = {v: ([], []) for v in variants}
        for r in range(6):
            for v in variants:  # interleaved: same process, same device, same data
                eng.search(q, k)
                info = eng.info()
                if r:
                    res[v][0].append(info["last_total_ms"])
                    res[v][1].append(info["last_scan_ms"])
        for v in variants:
            tot, scan = sorted(res[v][0]), sorted(res[v][1])
            print("Q=%d k=%d div=%d variant %s: pipeline %.3f ms (scan median %.3f min %.3f)  cand/query %.0f -> %.0f q/s"
                  % (nq, k, d, v, tot[2], scan[2], scan[0], info["last_candidates"] / nq, nq / tot[2] * 1e3),
                  flush=True)
