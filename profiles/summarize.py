#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-dispatch counters) for one kernel.

usage: profiles/summarize.py gpurun_out/prof_<tag> [kernel-substring]
Prints average duration per launch and the per-launch average of every counter collected.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel<0, false, 0>"
    out = {"kernel": needle}
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Name"]:
                out["calls"] = int(row["Calls"])
                out["avg_ns"] = float(row["AverageNs"])
                out["min_ns"] = float(row["MinNs"])
                out["max_ns"] = float(row["MaxNs"])
    counters = defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"]:
                counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                        "scratch": int(row["Scratch_Size"]), "grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"])}
    out["resources"] = meta
    out["counters_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
