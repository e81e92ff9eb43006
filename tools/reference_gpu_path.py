#!/usr/bin/env python
"""What the reference's own GPU search costs on this MI355X, beside ours, same data, same process.

A restatement in PyTorch-ROCm of DistributedBruteForceIndex.search_mips_index
(/root/reference/megatron/data/art_index.py:267-304) for ONE visible GPU: fp16 torch.matmul into a
[Q, N/G] slab per GPU (:278-280), copy of the slabs into a zero-filled [Q, N] fp16 matrix (:283-291),
torch.topk over it (:294), and the Python double loop that maps row positions to passage ids with one
.item() per element (:297-302).  Measurement infrastructure only (like bench.py's CPU baseline): the
product path never does any of this.  Prints one JSON line.
usage: reference_gpu_path.py [rows] [queries] [top_k]"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_WIKI
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)

# the same synthetic index as bench.build_shard, kept as one torch tensor for the reference path
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
E = torch.empty((n, 768), dtype=torch.float16, device=dev)
scale = 1.0 / math.sqrt(768)
for off in range(0, n, 1 << 18):
    m = min(1 << 18, n - off)
    E[off:off + m] = (torch.randn((m, 768), generator=gen, device=dev, dtype=torch.float32) * scale).half()
indices_arr = np.arange(1, n + 1, dtype=np.int64)
id_map = indices_arr  # position -> id; the reference builds a dict of the same content (:257-259)
gq = torch.Generator(device=dev)
gq.manual_seed(4321)
q = torch.randn((nq, 768), generator=gq, device=dev, dtype=torch.float32).half()


def reference_search(query_embeds, top_k, parts):
    t = [time.perf_counter()]
    C_ = torch.matmul(query_embeds, E.T)                                                   # :280
    torch.cuda.synchronize(); t.append(time.perf_counter())
    C = torch.zeros(query_embeds.shape[0], n, dtype=torch.float16, device="cuda")          # :283
    C[:, 0:n].copy_(C_)                                                                    # :291
    torch.cuda.synchronize(); t.append(time.perf_counter())
    distances, indices = torch.topk(C, top_k, dim=1)                                       # :294
    torch.cuda.synchronize(); t.append(time.perf_counter())
    fresh_indices = torch.zeros(indices.shape, dtype=torch.int32)                          # :297
    num_rows, num_cols = indices.shape
    for i in range(num_rows):                                                              # :299-301
        for j in range(num_cols):
            fresh_indices[i, j] = id_map[indices[i, j].item()]
    indices = fresh_indices.cuda()                                                         # :302
    torch.cuda.synchronize(); t.append(time.perf_counter())
    for name, a, b in zip(("matmul", "gather_slabs", "topk", "id_loop"), t[:-1], t[1:]):
        parts.setdefault(name, []).append(b - a)
    return distances, indices


parts = {}
reference_search(q, k, {})  # warm-up
times = []
for _ in range(5):
    t0 = time.perf_counter()
    d_ref, i_ref = reference_search(q, k, parts)
    times.append(time.perf_counter() - t0)
times.sort()
ref_s = times[len(times) // 2]

# ours on the same rows (uploaded from the same tensor), same queries
eng = MipsEngine(768, torch.float16, [0])
eng.begin(n)
ids32 = torch.arange(1, n + 1, device=dev, dtype=torch.int32)
for off in range(0, n, 1 << 20):
    m = min(1 << 20, n - off)
    eng.upload_device(0, E[off:off + m], ids32[off:off + m], off)
eng.finalize()
for _ in range(10):
    eng.search(q, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    d, i = eng.search(q, k)
torch.cuda.synchronize()
ours_s = (time.perf_counter() - t0) / 20

# agreement: same rows up to the reference's fp16 rounding of the scores before its top-k
same_sets = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i.cpu(), i_ref.cpu())) / float(nq * k)
exact_rows = float((i.cpu() == i_ref.cpu()).all(dim=1).float().mean())
# referee: float64 scores of the union of both answers.  Ours must be the k best of the union in
# (score desc, id asc) order; every pick of the reference that is not among them is a row it ranked
# too high after rounding its scores to fp16.
ours_wrong = 0
ref_not_topk = 0
worst_ref_rank = 0
max_rel = 0.0
for qi in range(nq):
    union = torch.unique(torch.cat([i[qi].long(), i_ref[qi].long()]))
    exact = (E[union - 1].double() @ q[qi].double())
    order = sorted(range(union.numel()), key=lambda t: (-float(exact[t]), int(union[t])))
    best = [int(union[t]) for t in order[:k]]
    ours_wrong += sum(1 for a, b in zip(best, i[qi].tolist()) if a != b)
    rank_of = {int(union[t]): r for r, t in enumerate(order)}
    ranks = [rank_of[int(x)] for x in i_ref[qi].tolist()]
    ref_not_topk += sum(1 for r in ranks if r >= k)
    worst_ref_rank = max(worst_ref_rank, max(ranks))
    top_exact = torch.tensor([float(exact[t]) for t in order[:k]], dtype=torch.float64)
    max_rel = max(max_rel, float(((d[qi].double().cpu() - top_exact).abs() / top_exact.abs()).max()))
print(json.dumps({
    "rows": n, "queries": nq, "top_k": k,
    "reference_gpu_path_ms": ref_s * 1e3, "reference_gpu_path_qps": nq / ref_s,
    "reference_parts_ms": {name: sorted(v)[len(v) // 2] * 1e3 for name, v in parts.items()},
    "ours_ms": ours_s * 1e3, "ours_qps": nq / ours_s, "speedup": ref_s / ours_s,
    "id_set_overlap": same_sets, "rows_identical_in_order": exact_rows,
    "referee_fp64_over_union": {"ours_positions_wrong": ours_wrong, "reference_picks_outside_true_topk": ref_not_topk,
                                "of": nq * k, "worst_true_rank_of_a_reference_pick": worst_ref_rank,
                                "ours_max_rel_score_error_fp16_output": max_rel},
    "note": "reference = art_index.py:267-304 restated with PyTorch-ROCm on one MI355X (fp16 matmul, full [Q,N] "
            "score matrix, torch.topk, Q*k .item() calls); differences in order come from its fp16 rounding of "
            "the scores before the top-k",
}))
eng.close()
