#!/usr/bin/env python3
"""Spilled-SGPR traffic INSIDE the stage loops of the kNN kernels, from device assembly (hipcc --cuda-device-only -S):
per kernel, the basic blocks that hold matrix instructions and, in them, the v_readlane / v_writelane (SGPR spill reloads / saves),
scratch loads / stores and s_load instructions.  The resource report only counts spilled registers; what costs time is a reload per stage.
usage: loop_spills.py file.s [more.s ...] [--write-baseline]   (--write-baseline: tests/golden/loop_spills.json, the guard of
tests/test_kernel_resources_cpu.py::test_stage_loops_hold_no_scratch_traffic_and_no_new_scalar_reloads)"""
import re, subprocess, sys

def kernels(path):
    cur, blocks, out = None, None, {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); blocks = out.setdefault(cur, [[]]); continue
        if cur is None:
            continue
        if re.match(r"^\s*\.end_amdhsa_kernel|^\s*s_endpgm", line):
            pass
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append([]); continue
        t = line.strip()
        if t and not t.startswith((".", ";")):
            blocks[-1].append(t.split()[0])
    return out

def summary(path):
    """{demangled kernel name: {"mfma", "readlane", "writelane", "scratch", "s_load"} over the basic blocks that hold >= 8 matrix instructions}"""
    ks = kernels(path)
    names = [k for k in ks if "knn" in k and "kernel" in k]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    out = {}
    for name, d in zip(names, dem):
        tot = {"mfma": 0, "readlane": 0, "writelane": 0, "scratch": 0, "s_load": 0}
        for b in ks[name]:
            n = sum(1 for i in b if i.startswith("v_mfma"))
            if n < 8:
                continue
            tot["mfma"] += n
            tot["readlane"] += sum(1 for i in b if i.startswith("v_readlane"))
            tot["writelane"] += sum(1 for i in b if i.startswith("v_writelane"))
            tot["scratch"] += sum(1 for i in b if i.startswith("scratch_"))
            tot["s_load"] += sum(1 for i in b if i.startswith("s_load"))
        out[d.replace("void ", "").split("(")[0]] = tot
    return out


if __name__ == "__main__":
    import json, os
    allk = {}
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for k, tot in summary(path).items():
            print(f"{k[:52]:52s} blocks with MFMAs: {tot}")
            allk[k] = tot
    if "--write-baseline" in sys.argv:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        json.dump(allk, open(os.path.join(root, "tests", "golden", "loop_spills.json"), "w"), indent=1, sort_keys=True)
        print("baseline written")
