#!/usr/bin/env python
"""Summarise rocprofv3 --pmc/--kernel-trace CSV directories: for the FILTER scan kernel (the dominant
kernel), per directory: launches, mean duration, mean counter values per launch, and the derived
figures DESIGN.md quotes (clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA busy = MFMA busy cycles /
(4 SIMDs x 256 CUs x cycles), wave-state fractions).  usage: pmc_summary.py DIR [DIR ...]"""
import csv
import glob
import json
import os
import re
import sys


def is_filter_scan(name):
    if "scan_kernel" not in name:
        return False
    m = re.search(r"scan_kernelI\w+?Li\d+ELi\d+ELi(\d+)E", name)
    mode = m.group(1) if m else name.split("<", 1)[1].split(",")[3].strip()
    return mode == "0"


out = {}
for d in sys.argv[1:]:
    cc = glob.glob(os.path.join(d, "*counter_collection.csv"))
    kt = glob.glob(os.path.join(d, "*kernel_trace.csv"))
    per = {}
    dur = {}
    if cc:
        with open(cc[0]) as f:
            for row in csv.DictReader(f):
                if not is_filter_scan(row["Kernel_Name"]):
                    continue
                k = int(row["Dispatch_Id"])
                per.setdefault(k, {})
                per[k][row["Counter_Name"]] = per[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[k] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    if kt:
        with open(kt[0]) as f:
            for row in csv.DictReader(f):
                if is_filter_scan(row["Kernel_Name"]):
                    dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    keys = sorted(per)
    # the first search of the process warms up: keep the launches of the last search only (a third)
    keep = keys[-max(1, len(keys) // 3):]
    n = len(keep)
    mean = {}
    for k in keep:
        for c, v in per[k].items():
            mean[c] = mean.get(c, 0.0) + v / n
    ms = sum(dur[k] for k in keep if k in dur) / max(1, sum(1 for k in keep if k in dur))
    o = {"launches_averaged": n, "launch_ms": ms, "counters_per_launch": mean}
    if "GRBM_GUI_ACTIVE" in mean and ms > 0:
        cyc = mean["GRBM_GUI_ACTIVE"] / 8.0
        o["clock_ghz"] = cyc / (ms * 1e-3) / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            o["mfma_busy_frac_of_simd_cycles"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    if "SQ_WAVE_CYCLES" in mean:
        w = mean["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in mean:
                o[c.lower() + "_frac"] = mean[c] / w
    if "FETCH_SIZE" in mean:
        o["hbm_fetch_bytes_per_launch_x2_corrected"] = mean["FETCH_SIZE"] * 1024 * 2
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
        o["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    out[os.path.basename(d.rstrip("/"))] = o
print(json.dumps(out, indent=1))
