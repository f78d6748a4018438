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
    needle = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel<0, 0, 0>"
    out = {"kernel": needle}
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Name"]:
                out["calls"] = int(row["Calls"])
                out["avg_ns"] = float(row["AverageNs"])
                out["min_ns"] = float(row["MinNs"])
                out["max_ns"] = float(row["MaxNs"])
    # The same launches one by one: bench.py's one-shot frame adds a short probe launch and its remainder to the
    # list, so the figure to hold against bench.py's kernel_ms is the mean over the full-length launches (within
    # 10 % of the median).
    durs = []
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"]:
                durs.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    if durs:
        durs.sort()
        med = durs[len(durs) // 2]
        full = [d for d in durs if abs(d - med) <= 0.1 * med]
        out["full_launches"] = len(full)
        out["full_launch_avg_ns"] = sum(full) / len(full)
    counters = defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"]:
                counters[row["Counter_Name"]].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
                meta = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                        "scratch": int(row["Scratch_Size"]), "grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"])}
    out["resources"] = meta
    # per-launch averages over the dispatches of the plan's usual size (a probe launch of a one-shot frame has the
    # built-in order's grid and a sixth of the samples: left out where it can be told apart, by its counter value)
    def usual(v):
        vals = sorted(x for _, x in v)
        med = vals[len(vals) // 2]
        keep = [x for x in vals if med == 0 or abs(x - med) <= 0.25 * abs(med)]
        return sum(keep) / len(keep)
    c = out["counters_per_launch"] = {k: usual(v) for k, v in sorted(counters.items())}
    # Derived figures.  The SQ counters tick in quad-cycles and are summed over the chip; GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs.  1024 SIMDs (256 CUs x 4).
    simds, xcds = 1024.0, 8.0
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / xcds                      # shader cycles the launch lasted
        d["kernel_cycles"] = round(cycles)
        if "SQ_WAVE_CYCLES" in c:
            d["mean_waves_per_simd"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (cycles * simds), 2)                   # of 8 slots
        if "SQ_INSTS_VALU" in c:
            d["cycles_per_valu_instruction_per_simd"] = round(cycles * simds / c["SQ_INSTS_VALU"], 2)           # 2 = the pipe's best
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_percent_rocprof_formula"] = round(100.0 * 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * simds), 1)
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_lane_utilisation_percent"] = round(100.0 * c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 1)   # 100 = no divergence
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["wave_time_waiting_for_issue_percent"] = round(100.0 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 1)
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_SALU" in c:
        d["salu_per_valu_instruction"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3)
    if "SQ_INSTS_VALU_TRANS" in c and "SQ_INSTS_VALU" in c:
        d["transcendental_share_of_valu_percent"] = round(100.0 * c["SQ_INSTS_VALU_TRANS"] / c["SQ_INSTS_VALU"], 2)
    if "SQ_LDS_IDX_ACTIVE" in c and "kernel_cycles" in d:
        # the LDS of a CU is busy while its index unit is: cycles summed over the 256 CUs
        d["lds_busy_percent"] = round(100.0 * c["SQ_LDS_IDX_ACTIVE"] / (d["kernel_cycles"] * 256.0), 1)
        if "SQ_LDS_BANK_CONFLICT" in c:
            d["lds_busy_cycles_lost_to_bank_conflicts_percent"] = round(100.0 * c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 1)
        if "SQ_INSTS_LDS" in c and c.get("SQ_WAVES"):
            d["lds_instructions_per_wave"] = round(c["SQ_INSTS_LDS"] / c["SQ_WAVES"], 1)
    if "cycles_per_valu_instruction_per_simd" in d:
        # what fills the launch: the share of the vector pipe's issue slots in use (2 cycles per instruction at best)
        # against the share of the time the LDS is busy
        valu = 2.0 / d["cycles_per_valu_instruction_per_simd"]
        d["valu_issue_slots_used"] = round(valu, 3)
        d["bound"] = "lds" if d.get("lds_busy_percent", 0.0) / 100.0 > valu else "valu"
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes_per_launch"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)                   # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    out["derived"] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
