#!/usr/bin/env python
"""A/B of pass 0 keeping one vs four sample values per lane (ARTMIPS_TOPT), large k, same process:
C5 shape (8,841,823 rows) and a 1/8 Wikipedia shard (2,626,916 rows), 2,048 queries, top-1000, bf16."""
import json
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

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4321)
for rows in (8841823, 2626916):
    eng = MipsEngine(768, torch.bfloat16, [0])
    bench.build_shard(eng, 0, rows, 1234, torch.bfloat16)
    q = torch.randn((2048, 768), generator=g, device=dev).bfloat16()
    ref = None
    for rep in range(2):
        for topt in ("1", "4"):
            os.environ["ARTMIPS_TOPT"] = topt
            eng.search(q, 1000)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                s, i = eng.search(q, 1000)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            info = eng.info()
            if ref is None:
                ref = i.clone()
            print(json.dumps({"rows": rows, "topt": int(topt), "ms_per_call": dt * 1e3, "ms_per_sweep": dt * 1e3 / 8,
                              "candidates_per_query": info["last_candidates"] / 2048, "retries": info["last_retries"],
                              "ids_equal_first": bool(torch.equal(i, ref))}), flush=True)
    eng.close()
