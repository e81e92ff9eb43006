#!/usr/bin/env python
"""GPU timeline of the last few steps of a bench run traced with
rocprofv3 --kernel-trace --memory-copy-trace: start offset, gap to the previous operation and duration of
every kernel / copy of a step (a step starts at each pass-0 scan).  usage: step_timeline.py DIR [n_steps]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops = []
for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"scan_kernelI\w+?Li\d+ELi\d+ELi(\d+)E", name)
        if m:
            name = "scan_kernel<%s>" % ("FILTER" if m.group(1) == "0" else "GROUPMAX (pass 0)")
        else:
            name = re.sub(r"\(.*", "", name).replace("artk::", "").replace("void ", "")[:60]
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(os.path.join(d, "*memory_copy_trace.csv")):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ops.sort()
starts = [i for i, o in enumerate(ops) if "pass 0" in o[2]]
for si in starts[-n_steps - 1:-1]:
    nxt = starts[starts.index(si) + 1]
    t0 = ops[si][0]
    print("step: %.1f us from first kernel to the next step's first kernel" % ((ops[nxt][0] - t0) / 1e3))
    prev_end = t0
    for s, e, name in ops[si:nxt]:
        print("  + %7.1f us  gap %5.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
        prev_end = max(prev_end, e)
    print("  GPU idle until the next step's first kernel: %.1f us" % ((ops[nxt][0] - prev_end) / 1e3))
