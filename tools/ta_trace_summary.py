#!/usr/bin/env python3
"""Joins tools/ta_profile.py's launch order with a rocprofv3 kernel trace CSV: per shape the mean kernel-only duration of
tree_attention_kernel (+ merge kernel when the keys were split), achieved GB/s and the HBM fraction.
Usage: python tools/ta_trace_summary.py <kernel_trace.csv> <ta_profile stdout> [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tree_attention" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "merge" not in r["Kernel_Name"]]
merge = [r for r in rows if "merge" in r["Kernel_Name"]]
out, mi, gi = [], 0, 0
for line in open(sys.argv[2]):
    if not line.startswith("SHAPE"):
        continue
    f = dict(kv.split("=") for kv in line.split()[2:])
    label, n, nbytes = line.split()[1], int(f["launches"]), int(f["bytes"])
    mains = main[mi:mi + n]
    mi += n
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mains][5:]
    # a merge launch belongs to the main launch it follows
    merged = []
    while gi < len(merge) and int(merge[gi]["Start_Timestamp"]) < (int(main[mi]["Start_Timestamp"]) if mi < len(main) else 1 << 62):
        merged.append(int(merge[gi]["End_Timestamp"]) - int(merge[gi]["Start_Timestamp"]))
        gi += 1
    us = sum(dur) / len(dur) / 1e3
    mus = sum(merged[5:]) / max(len(merged[5:]), 1) / 1e3 if merged else 0.0
    out.append({"shape": label, "kernel": mains[0]["Kernel_Name"].split("(")[0], "grid": mains[0].get("Grid_Size_X", ""), "main_us": round(us, 2),
                "merge_us": round(mus, 2), "GBps": round(nbytes / (us + mus) / 1e3, 1), "frac_hbm": round(nbytes / (us + mus) / 1e3 / 8000, 3),
                "TFLOPs": round(float(f["flops"]) / (us + mus) / 1e6, 1)})
    print(out[-1])
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
