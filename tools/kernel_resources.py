#!/usr/bin/env python3
"""Registers / spills / scratch of every kNN kernel, from the resource reports the build leaves in open-hummingbird-eval_amd/lib/
(untracked lib/build/: csrc/Makefile compiles the kNN units with -Rpass-analysis=kernel-resource-usage).

usage: kernel_resources.py                  print one line per kernel
       kernel_resources.py --write-baseline  also rewrite tests/golden/kernel_resources.json (the spill guard's baseline:
                                             tests/test_kernel_resources_cpu.py fails when a shipped kernel spills more than this)"""
import glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "open-hummingbird-eval_amd", "lib", "build")
FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "waves_per_simd"}


def parse(lib_dir=LIB):
    rows = {}
    for f in sorted(glob.glob(os.path.join(lib_dir, "resources_*.txt"))):
        cur = None
        for line in open(f, errors="replace"):
            m = re.search(r"remark: .*?Function Name: (\S+)", line)
            if m:
                cur = m.group(1); rows[cur] = {"unit": os.path.basename(f)[len("resources_"):-4]}
                continue
            m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
            if m and cur:
                rows[cur][FIELDS[m.group(1)]] = int(m.group(2))
    names = list(rows)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    return {d.replace("void ", "").split("(")[0]: rows[n] for d, n in zip(dem, names)}


if __name__ == "__main__":
    r = parse()
    for k, d in r.items():
        print(f"{k[:64]:64s} {d['unit']:14s} VGPR {d.get('vgpr', 0):3d} spill S {d.get('sgpr_spill', 0):3d} V {d.get('vgpr_spill', 0):3d} "
              f"scratch {d.get('scratch', 0):4d} B  waves/SIMD {d.get('waves_per_simd', 0)}")
    if "--write-baseline" in sys.argv:
        json.dump(r, open(os.path.join(ROOT, "tests", "golden", "kernel_resources.json"), "w"), indent=1, sort_keys=True)
        print("baseline written")
