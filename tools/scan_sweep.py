#!/usr/bin/env python
"""A/B settings of the scan in ONE process on ONE index, interleaved rounds, median and min of the
HIP-event time of the FILTER scan per variant.  Variants are values of an environment knob of the
library: --nq-sweep (ARTMIPS_MAX_NQ), --shape-sweep (ARTMIPS_NW:ARTMIPS_MAX_NQ[:ARTMIPS_ABLATE]) or
--env NAME.  (The kernel builds compared in profiles/r01_b_scan_sweep.txt -- LDS image, cache policy,
ring depth -- were removed from the library once the shipped configuration was chosen.)"""
import argparse
import math
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

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=bench.N_WIKI)
p.add_argument("--variants", required=True)
p.add_argument("--rounds", type=int, default=6)
p.add_argument("--queries", type=int, default=64)
p.add_argument("--topk", type=int, default=32)
p.add_argument("--nq-sweep", action="store_true", help="variants are values of ARTMIPS_MAX_NQ (1..4)")
p.add_argument("--env", default=None, help="variants are values of this environment variable")
p.add_argument("--envs", action="store_true", help="variants are NAME=VAL+NAME=VAL settings of several knobs")
p.add_argument("--shape-sweep", action="store_true", help="variants are NW:NQ pairs (ARTMIPS_NW, ARTMIPS_MAX_NQ)")
args = p.parse_args()

eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, args.rows, 1234, torch.float16)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((args.queries, 768), generator=g, device="cuda").half()
variants = [v for v in args.variants.split(",")]
times = {v: [] for v in variants}
totals = {v: [] for v in variants}
ref = None
for rnd in range(args.rounds + 1):
    for v in variants:
        if args.nq_sweep:
            os.environ["ARTMIPS_MAX_NQ"] = v
        if args.env:
            os.environ[args.env] = v
        if args.envs:
            for kv in v.split("+"):
                name, val = kv.split("=")
                os.environ[name] = val
        if args.shape_sweep:
            os.environ["ARTMIPS_NW"], os.environ["ARTMIPS_MAX_NQ"] = v.split(":")[:2]
            os.environ["ARTMIPS_ABLATE"] = v.split(":")[2] if v.count(":") > 1 else "0"
        s, i = eng.search(q, args.topk, score_dtype=torch.float32)
        info = eng.info()
        if rnd > 0:
            times[v].append(info["last_scan_ms"])
            totals[v].append(info["last_total_ms"])
        if os.environ.get("ARTMIPS_ABLATE", "0") != "0":
            pass  # ablated runs compute nothing meaningful
        elif ref is None:
            ref = (s.clone(), i.clone())
        else:
            same = bool((i == ref[1]).all())
            close = bool(torch.allclose(s, ref[0], rtol=1e-4, atol=1e-5))
            if not (same and close):
                print("variant %s: ids equal %s, scores close %s  <-- MISMATCH" % (v, same, close), flush=True)
bytes_ = args.rows * 768 * 2
for v in variants:
    t = sorted(times[v])
    tt = sorted(totals[v])
    med, mn = t[len(t) // 2], t[0]
    tot = tt[len(tt) // 2]
    print("variant %s: scan median %.3f ms (%.0f GB/s per sweep-byte)  min %.3f ms  pipeline median %.3f ms  "
          "-> %.0f queries/s, %.1f TFLOP/s"
          % (v, med, bytes_ * info["last_scan_launches"] / med / 1e6, mn, tot, args.queries / tot * 1e3,
             2.0 * args.queries * args.rows * 768 / tot / 1e9), flush=True)
