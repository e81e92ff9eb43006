#!/usr/bin/env python
"""Rate of the way back to the host (artmips_download): a 4M-row fp16 shard read back in 128K-row
chunks, alone and while a second thread keeps searching the same index."""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, n, 1234, torch.float16)
q = torch.randn((64, 768), device="cuda").half()
eng.search(q, 32)


def pull():
    gen = eng.generation()
    t0 = time.perf_counter()
    for lo in range(0, n, 1 << 17):
        eng.download(lo, min(1 << 17, n - lo), gen)
    return time.perf_counter() - t0


pull()
alone = pull()
stop = False
count = [0]


def searcher():
    while not stop:
        eng.search(q, 32)
        count[0] += 1


t0 = time.perf_counter()
for _ in range(50):
    eng.search(q, 32)
ms_alone = (time.perf_counter() - t0) / 50 * 1e3
th = threading.Thread(target=searcher)
th.start()
t0 = time.perf_counter()
beside = pull()
dt = time.perf_counter() - t0
stop = True
th.join()
gb = n * 1540 / 1e9
print(json.dumps({"rows": n, "download_GBs_alone": gb / alone, "download_GBs_beside_searches": gb / beside,
                  "search_ms_alone": ms_alone, "search_ms_beside_download": dt / max(1, count[0]) * 1e3,
                  "wikipedia_scale_seconds_est": 21015324 * 1540 / 1e9 / (gb / alone)}))
eng.close()
