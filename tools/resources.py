#!/usr/bin/env python3
"""Table of the kernels' register / LDS / scratch usage from `make -C physics-based-ray-tracing_amd/csrc resources`
(-Rpass-analysis=kernel-resource-usage).  Usage: tools/resources.py [substring]"""
import re
import subprocess
import sys

out = subprocess.run(["make", "-C", "physics-based-ray-tracing_amd/csrc", "resources"], capture_output=True, text=True).stderr
want = sys.argv[1] if len(sys.argv) > 1 else "k_"
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(RadArgs)", "").replace("(UsArgs)", "")
    if want in name:
        print(f"{name[:60]:60s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} "
              f"{r.get('SGPRs Spill', 0):6d} {r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):4d} {r.get('LDS Size', 0):7d}")
