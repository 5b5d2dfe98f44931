#!/usr/bin/env python3
"""Registers / spills / scratch of every kernel of a .hip unit (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: kernel_resources.py unit.hip [extra hipcc flags...]   (run in csrc/)"""
import re, subprocess, sys
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      "-c", sys.argv[1], "-o", "/dev/null"] + sys.argv[2:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    print(f"{name[:70]:70s} VGPR {d.get('VGPRs', 0):3d} AGPR {d.get('AGPRs', 0):3d} spill S {d.get('SGPRs Spill', 0):3d} V {d.get('VGPRs Spill', 0):3d} scratch {d.get('ScratchSize', 0):4d} B")
