#!/usr/bin/env python
"""Randomised parity stress: many (rows, queries, k, dim, dtype, shards) shapes against the oracle.
Not part of the test suite (minutes of CPU oracle time); run on the GPU box:  stress.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd.engine import MipsEngine  # noqa: E402
from oracle import mips_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.Generator(np.random.PCG64(seed))
t_start = time.time()
bad = 0
for c in range(cases):
    dim = int(rng.choice([128, 768, 768, 768, 1024, 256, 384, 512, 640, 896]))
    storage = str(rng.choice(["f16", "f16", "bf16", "f32"]))
    if storage == "f32" and dim not in (128, 768):
        dim = 768
    big = os.environ.get("STRESS_BIG") == "1"  # larger shards, multi-sweep query counts
    n = int(np.exp(rng.uniform(np.log(2000 if big else 1), np.log(400000 if big else 150000))))
    qmax = max(1, min(700, int((1.5e8 if big else 2.0e7) / max(n, 1) / (dim / 128))))
    q = int(np.exp(rng.uniform(np.log(min(qmax, 60)) if big else 0, np.log(qmax))))
    k = int(np.exp(rng.uniform(0, np.log(min(n, 1500)))))
    shards = int(rng.choice([1, 1, 1, 2, 3, 5]))
    if storage == "f32":
        shards = 1 if q > 64 else shards
    ints = rng.random() < 0.25  # integer-valued data: many exact ties
    if ints:
        rows = rng.integers(-4, 5, size=(n, dim)).astype(np.float32)
        queries = rng.integers(-4, 5, size=(q, dim)).astype(np.float32)
    else:
        rows = (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
        queries = rng.standard_normal((q, dim)).astype(np.float32)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[storage]
    rs, qs = orc.to_storage(rows, storage), orc.to_storage(queries, storage)
    tag = "case %3d: n=%6d q=%3d k=%4d dim=%4d %s shards=%d %s" % (c, n, q, k, dim, storage, shards,
                                                                    "ints" if ints else "")
    try:
        eng = MipsEngine(dim, tdt, [0] * shards)
        eng.begin(n)
        eng.upload_host(torch.from_numpy(rows).to(tdt) if storage == "bf16" else rows.astype(
            np.float16 if storage == "f16" else np.float32), ids)
        eng.finalize()
        s, i = eng.search(torch.from_numpy(queries).to(tdt).cuda(), k, score_dtype=torch.float32)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        if ints:
            es, ei, _ = orc.exact_topk(qs, rs, ids, k)
            ok = np.array_equal(i, ei) and np.array_equal(s, es)
            msg = "" if ok else "integer case not bit-exact"
        else:
            nd = orc.check_topk(s, i, qs, rs, ids, k, rel_tol=1e-3)
            ok = nd <= max(2, (q * k) // 300)
            msg = "" if ok else "%d positions differ" % nd
        eng.close()
    except Exception as e:  # noqa: BLE001
        ok, msg = False, "%s: %s" % (type(e).__name__, e)
    if not ok:
        bad += 1
        print("FAIL " + tag + " -> " + msg, flush=True)
    elif c % 10 == 0:
        print("ok   " + tag + "  (%.0f s)" % (time.time() - t_start), flush=True)
print("%d cases, %d failures, %.0f s" % (cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
