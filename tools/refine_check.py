#!/usr/bin/env python
"""Two-part FILTER scan A/B in the diagnostic build: 2,048 queries x top-k over a synthetic shard with
ARTMIPS_REFINE = 0 (one part, fitted sample), 4 and 8 (every 4th / 8th tile first, 4 x coarser sample).
Prints candidates per query, the FILTER scan time and the whole pipeline time (HIP events) and checks
that the three settings return identical ids.  usage: refine_check.py ROWS [K] [QUERIES] [REFINE[:SAMPLE_DIV],...]"""
import os
import sys

import torch

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("ARTMIPS_DIAGNOSTICS", "1")
os.environ.setdefault("ARTMIPS_LIBRARY", R + "/art_amd/lib/libartmips_diag.so")   # (another diagnostic build may be named)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.float16)
q = torch.randn((nq, 768), device="cuda").half()
ref = None
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["0", "4", "8", "0", "8"]
for v in variants:   # REFINE or REFINE:SAMPLE_DIV
    v, _, nw = v.partition("/")          # REFINE[:SAMPLE_DIV][/NW]  (NW: ARTMIPS_NW, 44 = 4 x 4 gangs, 88 = 8 x 2)
    os.environ.pop("ARTMIPS_NW", None)
    if nw:
        os.environ["ARTMIPS_NW"] = nw
    r, _, sd = v.partition(":")
    os.environ["ARTMIPS_REFINE"] = r
    os.environ.pop("ARTMIPS_SAMPLE_DIV", None)
    if sd:
        os.environ["ARTMIPS_SAMPLE_DIV"] = sd
    for _ in range(2):
        s, ids = eng.search(q, k)
    i = eng.info()
    same = True if ref is None else bool(torch.equal(ids, ref))
    if ref is None:
        ref = ids.clone()
    print("refine", v, "nw", nw or "-", "cand/query %.0f" % (i["last_candidates"] / nq), "scan ms %.3f" % i["last_scan_ms"],
          "pipeline ms %.3f" % i["last_total_ms"], "ids identical", same, flush=True)
    if not same:
        sys.exit(1)
