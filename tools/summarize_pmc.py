#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel -> JSON summary.

usage: summarize_pmc.py <gpurun_out/TAG> <profiles/out.json>
Units: FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3).  Per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts
128-B requests at 64 B, i.e. reports half of a wide streaming read: the summary
carries both the raw figure and the x2-corrected one.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            a = agg[k][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {}
    for k, counters in agg.items():
        rec = {}
        for c, (n, tot) in counters.items():
            rec[c] = {"dispatches": n, "mean": tot / n}
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            f = rec["FETCH_SIZE"]["mean"] * 1024
            w = rec["WRITE_SIZE"]["mean"] * 1024
            rec["hbm_bytes_per_launch_raw"] = f + w
            rec["hbm_bytes_per_launch"] = 2 * f + w     # gfx950 FETCH_SIZE x2 correction
        out[k] = rec
    stats = os.path.join(src, "kt", "kt_kernel_stats.csv")
    if os.path.exists(stats):
        out["_kernel_stats"] = [dict(r) for r in csv.DictReader(open(stats))]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1)[:3000])


if __name__ == "__main__":
    main()
