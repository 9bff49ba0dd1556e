#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/occupancy per kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["hipcc", "-c", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "ros_gpu_stereo_processor_amd", "csrc"), src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
for r in rows:
    print("%-70s vgpr %4s sgpr %4s scratch %5s occ %2s" % (r["name"][:70], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
