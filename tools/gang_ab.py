#!/usr/bin/env python
"""A/B of scan-kernel code variants inside ONE process on ONE index (diagnostic build): the variants are
selected per call through ARTMIPS_ABLATE (bits 64 / 128 / 256, artmips_kernels.h ScanArgs::ablate; their
results are identical, which is checked), interleaved round by round so that clock drift and the box
drawn hit all arms alike.

usage: ARTMIPS_DIAGNOSTICS=1 ARTMIPS_LIBRARY=.../libartmips_diag.so python tools/gang_ab.py ROWS Q K DTYPE ROUNDS ARM [ARM ...]
       an ARM is an ARTMIPS_ABLATE value, optionally name=value; value/shift/lead also sets ARTMIPS_THR_SHIFT (1/100) and ARTMIPS_SYNC_LEAD
prints per arm: median / min of the summed scan-kernel time of a call (HIP events) and of the whole call.
"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from art_amd.engine import MipsEngine  # noqa: E402

rows, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[4]]
rounds = int(sys.argv[5])
arms = []
for a in sys.argv[6:]:
    name, _, val = a.rpartition("=")
    arms.append((name or ("a" + val), val))

eng = MipsEngine(768, dt, [0])
bench.build_shard(eng, 0, rows, 1234, dt)
g = torch.Generator(device="cuda")
g.manual_seed(4321)
q = torch.randn((nq, 768), generator=g, device="cuda").to(dt)
time.sleep(float(os.environ.get("GANG_AB_SETTLE", "6")))

ref = None
scan = {n: [] for n, _ in arms}
wall = {n: [] for n, _ in arms}
def select(val):
    parts = (val.split("/") + ["", ""])[:3]
    os.environ["ARTMIPS_ABLATE"] = parts[0]
    os.environ["ARTMIPS_THR_SHIFT"] = parts[1] or "0"
    os.environ["ARTMIPS_SYNC_LEAD"] = parts[2] or "0"


cands = {}
for name, val in arms:  # warm every arm (first-use attribute calls, buffer growth, clock ramp)
    select(val)
    for _ in range(2):
        s, i = eng.search(q, k)
    if ref is None:
        ref = i.clone()
    else:
        # (GANG_AB_NOCHECK=1: arms that switch parts of the loop off, bits 1..32, return garbage by design)
        assert os.environ.get("GANG_AB_NOCHECK") or torch.equal(ref, i), \
            "arm %s returns other ids than arm %s" % (name, arms[0][0])
for r in range(rounds):
    order = arms if r % 2 == 0 else arms[::-1]
    for name, val in order:
        select(val)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.search(q, k)
        torch.cuda.synchronize()
        wall[name].append((time.perf_counter() - t0) * 1e3)
        scan[name].append(eng.info()["last_scan_ms"])
        cands[name] = eng.info()["last_candidates"]
info = eng.info()
print("rows %d Q %d k %d %s, %d scan launches per call, %d rounds, %d candidates kept by the last call's filter" % (
    rows, nq, k, sys.argv[4], info["last_scan_launches"], rounds, info["last_candidates"]))
for name, val in arms:
    print("  %-22s ablate %-4s scan median %.3f min %.3f ms | call median %.3f min %.3f ms | %d candidates" % (
        name, val, statistics.median(scan[name]), min(scan[name]), statistics.median(wall[name]), min(wall[name]),
        cands[name]))
eng.close()
