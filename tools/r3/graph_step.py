#!/usr/bin/env python
"""1/8-shard training step (64 x top-32 over 2,626,916 rows): the synchronous entry point against a
captured HIP graph of the allocation-free entry point (artmips_search_local_noalloc) + the packed merge,
replayed per step.  Prints ms per step for both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2626916
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, rows, 1234, torch.float16)
g = torch.Generator(device=dev)
g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device=dev).half()
k = 32
eng.set_stats(False)
time.sleep(2)


def timed(fn, n):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def sync_step():
    return eng.search(q, k)


words = MipsEngine.packed_words(64, k)
packed = torch.zeros((1, words), dtype=torch.int32, device=dev)
flag = torch.zeros((1,), dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eng.search_local_noalloc(0, q, k, packed[0])
    out = MipsEngine.merge_packed(packed, 64, k, out_flag=flag)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    eng.search_local_noalloc(0, q, k, packed[0])
    gs, gi = MipsEngine.merge_packed(packed, 64, k, out_flag=flag)


def graph_step():
    graph.replay()
    torch.cuda.current_stream().synchronize()


def noalloc_step():
    eng.search_local_noalloc(0, q, k, packed[0])
    s, i = MipsEngine.merge_packed(packed, 64, k, out_flag=flag)
    torch.cuda.current_stream().synchronize()
    return s, i


ws, wi = sync_step()
graph_step()
assert torch.equal(gi, wi) and torch.equal(gs, ws), "graph replay differs from the synchronous search"
for rep in range(2):
    print("rows %d: sync search %.4f ms | noalloc + merge, enqueued per step %.4f ms | graph replay %.4f ms per step"
          % (rows, timed(sync_step, steps), timed(noalloc_step, steps), timed(graph_step, steps)), flush=True)
eng.close()
