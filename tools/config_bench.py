#!/usr/bin/env python
"""Throughput of the other BASELINE.json configs on ONE GPU (they are parity-test shapes, not the
headline bench line): C3 NQ-Open eval (3,610 queries, top-100; one call and the reference's 16-query
chunking), C4 large batch (16,384 queries, top-1000, bf16), C5 MS MARCO scale (8,841,823 rows, 6,980
queries, top-1000, bf16).  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c3", "c4", "c5"]
dev = torch.device("cuda", 0)


def run(tag, eng, q, k, chunk=None, reps=2, elem=2, peak=2500.0, per_sweep=256):
    n_rows = eng.n_rows
    def once():
        if chunk is None:
            return eng.search(q, k)
        outs = [eng.search(q[i:i + chunk], k) for i in range(0, q.shape[0], chunk)]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s, i = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    info = eng.info()
    flops = 2.0 * q.shape[0] * n_rows * 768
    roof = None
    if chunk is None:
        # SURVEY.md §8(d), MFMA-bound regime: both fractions for the scan kernels of the last call
        # (HIP events on the launch stream, summed over the sweeps): flops / t against the dense MFMA
        # peak, and the HBM traffic of ceil(Q/256) sweeps against the HBM peak
        scan_s = info["last_scan_ms"] * 1e-3
        sweeps = info["last_scan_launches"]
        roof = {"bound": "mfma", "kernel": "scan_kernel<FILTER>, %d sweeps" % sweeps,
                "kernel_s_total": scan_s, "achieved": flops / scan_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flops / scan_s / (peak * 1e12),
                "hbm_achieved_GBs": sweeps * n_rows * 768 * elem / scan_s / 1e9,
                "hbm_frac": sweeps * n_rows * 768 * elem / scan_s / 8e12,
                "scan_share_of_call": scan_s / dt}
    print(json.dumps({"config": tag, "queries": q.shape[0], "top_k": k, "rows": n_rows, "chunk": chunk,
                      "seconds_per_pass": dt, "queries_per_s": q.shape[0] / dt, "tflops": flops / dt / 1e12,
                      "frac_of_mfma_peak": flops / dt / (peak * 1e12), "roofline": roof,
                      "index_sweeps": info["last_scan_launches"] if chunk is None else None,
                      "candidates_per_query": info["last_candidates"] / (q.shape[0] if chunk is None else min(chunk, q.shape[0])),
                      "retries": info["last_retries"]}), flush=True)


g = torch.Generator(device=dev)
g.manual_seed(4321)
if "c3" in which:
    eng = MipsEngine(768, torch.float16, [0])
    bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.float16)
    q = torch.randn((3610, 768), generator=g, device=dev).half()
    run("C3 one call", eng, q, 100)
    run("C3 chunked like the reference (--shard-size 16)", eng, q, 100, chunk=16, reps=1)
    eng.close()
if "c4" in which:
    eng = MipsEngine(768, torch.bfloat16, [0])
    bench.build_shard(eng, 0, bench.N_WIKI, 1234, torch.bfloat16)
    q = torch.randn((16384, 768), generator=g, device=dev).bfloat16()
    run("C4 large batch bf16", eng, q, 1000, reps=1)
    eng.close()
if "c5" in which:
    eng = MipsEngine(768, torch.bfloat16, [0])
    bench.build_shard(eng, 0, 8841823, 1234, torch.bfloat16)
    q = torch.randn((6980, 768), generator=g, device=dev).bfloat16()
    run("C5 MS MARCO scale bf16", eng, q, 1000, reps=1)
    eng.close()
if "c5f32" in which:
    # BASELINE config 5's other half: the same shape with fp32 storage (exact-f32 MFMA, 157.3 TFLOP/s
    # dense peak, 64 queries per sweep: MI355X_MICROARCH.md, Matrix cores)
    eng = MipsEngine(768, torch.float32, [0])
    bench.build_shard(eng, 0, 8841823, 1234, torch.float32)
    q = torch.randn((6980, 768), generator=g, device=dev)
    run("C5 MS MARCO scale fp32 storage", eng, q, 1000, reps=1, elem=4, peak=157.3)
    eng.close()
