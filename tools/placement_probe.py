#!/usr/bin/env python
"""Does the scan rate depend on WHERE in HBM the index landed, and does the tile order matter then?
Builds the 21M-row index in fresh memory, frees it, rebuilds it in recycled memory, and sweeps
ARTMIPS_RUN_SHIFT (runs of 2^s consecutive tiles per workgroup) on both, interleaved."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device="cuda").half()
shifts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "3", "4", "6", "8", "12"]


def sweep(eng, tag):
    res = {s: [] for s in shifts}
    ref = None
    for rnd in range(7):
        for s in shifts:
            os.environ["ARTMIPS_RUN_SHIFT"] = s
            sc, ids = eng.search(q, 32)
            if ref is None:
                ref = ids.clone()
            assert torch.equal(ids, ref), "tile order changed the result"
            if rnd:
                res[s].append(eng.info()["last_scan_ms"])
    print(tag + ": " + "  ".join("s=%s %.3f" % (s, sorted(res[s])[len(res[s]) // 2]) for s in shifts), flush=True)


held = []
for trial in range(2):
    eng = MipsEngine(768, torch.float16, [0])
    bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.float16)
    sweep(eng, "fresh memory, build %d   " % trial)
    held.append(eng)
for eng in held:
    eng.close()
import time
for trial in range(4):
    if trial == 2:
        time.sleep(25)  # is the penalty a background activity that ends (page clearing), or the pages?
    eng = MipsEngine(768, torch.float16, [0])
    bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.float16)
    sweep(eng, "recycled memory, build %d%s" % (trial, " (after 25 s idle)" if trial >= 2 else ""))
    eng.close()
