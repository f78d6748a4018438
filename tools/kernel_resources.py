"""Registers, spills and scratch of every kernel instance, from hipcc's resource-usage remarks.
usage: python tools/kernel_resources.py   (compiles ray_tracing_amd/csrc/rt_kernels.hip for the device only)"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(root, "ray_tracing_amd", "csrc"), "asm"], capture_output=True, text=True)
os.remove(os.path.join(root, "ray_tracing_amd", "csrc", "rt_kernels.s"))
rows, cur = [], {}
for line in out.stderr.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = re.sub(r"_ZN3rtd12trace_kernelILi(\d)ELi(\d)ELi(\d)E.*", r"trace_kernel<\1,\2,\3>", r["name"])
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    print(f"{name[:40]:40s} vgpr {r.get('VGPRs'):>4s} sgpr {r.get('TotalSGPRs'):>4s} scratch {r.get('ScratchSize'):>4s} "
          f"vspill {r.get('VGPRs Spill'):>3s} sspill {r.get('SGPRs Spill'):>3s} occ {r.get('Occupancy')}")
