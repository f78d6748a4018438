#!/usr/bin/env python3
"""Files one profile run (gpurun_out/prof_<tag>, made by run_profile.sh) under profiles/: the kernel-stats CSV, the
summary (summarize.py) and the configuration's entry in traffic.json, which bench.py replays for roofline.traffic and
roofline.valu_issue together with the hash of the kernel sources the profile was taken with.

    python profiles/collect.py <run tag> <name under profiles/> <traffic.json key> "<kernel substring>"
    e.g. python profiles/collect.py r02a_c2 r02_c2 c2_1920x1080_100spp "trace_kernel<0, 0, 0>"
"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def kernel_source_sha():
    h = hashlib.sha256()
    for f in ("rt_kernels.hip", "rt_device.h"):
        h.update(open(os.path.join(ROOT, "ray_tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    tag, name, key, needle = sys.argv[1:5]
    run = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    stats = glob.glob(os.path.join(run, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(HERE, name + "_kernel_stats.csv"))
    out = subprocess.run([sys.executable, os.path.join(HERE, "summarize.py"), run, needle], check=True, capture_output=True, text=True).stdout
    summary = json.loads(out)
    if not stats or not summary.get("counters_per_launch"):
        sys.exit(f"{run}: no kernel trace or no counters there; nothing filed")
    summary["kernel_source_sha"] = kernel_source_sha()
    json.dump(summary, open(os.path.join(HERE, name + "_summary.json"), "w"), indent=1)
    c, d = summary["counters_per_launch"], summary["derived"]
    tpath = os.path.join(HERE, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic[key] = {
        "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
        "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"),
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts half the bytes of 16-B-per-lane coalesced reads on gfx950",
        "valu_instructions_per_launch": round(c["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in c else None,
        "valu_lane_utilisation_percent": d.get("valu_lane_utilisation_percent"),
        "mean_waves_per_simd": d.get("mean_waves_per_simd"),
        "lds_busy_percent": d.get("lds_busy_percent"),
        "bound": d.get("bound", "valu"),
        "kernel_launch_avg_ns_rocprof": summary.get("full_launch_avg_ns"),
        "kernel_sha": summary["kernel_source_sha"],
        "source": f"profiles/{name}_summary.json",
    }
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(name, "launch", round(summary.get("full_launch_avg_ns", 0) / 1e6, 4), "ms;", json.dumps(d))


if __name__ == "__main__":
    main()
