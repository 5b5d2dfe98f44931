"""Spill guard: every shipped kNN kernel instantiation is held to the committed register / spill / scratch baseline.
No test fails when a header change for ONE kernel pushes ANOTHER over its scalar registers -- it only gets slower (round 2: +12 % at
k = 90 from 16 reloads of spilled SGPRs per stage) -- so the build leaves hipcc's per-kernel resource report in lib/build/ (untracked) and this test
compares it with tests/golden/kernel_resources.json (rewrite it with `python tools/kernel_resources.py --write-baseline` when a
change is meant to move the numbers)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))),
                    reason="needs hipcc: the resource report is a by-product of compiling the kNN units")
def test_no_kernel_spills_more_than_the_baseline():
    subprocess.run(["make", "-C", os.path.join(ROOT, "open-hummingbird-eval_amd", "csrc"), "-j", "8"], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)          # a no-op when the library is up to date
    import kernel_resources
    now = kernel_resources.parse()
    base = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_resources.json")))
    knn = [k for k in now if k.startswith("knn_")]
    assert len(knn) >= 12, f"resource reports missing or unparsed: {sorted(now)}"
    problems = []
    for k, d in now.items():
        if not k.startswith("knn_"):
            continue
        if k not in base:
            problems.append(f"{k}: not in the baseline (python tools/kernel_resources.py --write-baseline)")
            continue
        b = base[k]
        if d.get("vgpr", 0) > 256:
            problems.append(f"{k}: {d['vgpr']} VGPRs -- one wave per SIMD")
        if d.get("waves_per_simd", 2) < b.get("waves_per_simd", 2):
            problems.append(f"{k}: occupancy {d.get('waves_per_simd')} < {b.get('waves_per_simd')} waves per SIMD")
        for f in ("sgpr_spill", "vgpr_spill", "scratch"):
            if d.get(f, 0) > b.get(f, 0):
                problems.append(f"{k}: {f} {d.get(f, 0)} > baseline {b.get(f, 0)}")
    assert not problems, "\n".join(problems)
    # the two kernels the bench times keep their stage loops free of scratch traffic beyond these few bytes (prologue / segment ends)
    assert now["knn_fused_bd_kernel<false, true, false>"]["scratch"] <= 36 and now["knn_f16v2_kernel<4>"]["scratch"] == 0


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))), reason="needs hipcc")
def test_stage_loops_hold_no_scratch_traffic_and_no_new_scalar_reloads(tmp_path):
    """The resource report counts spilled registers; what costs time -- and, for scratch, CORRECTNESS: the stage loops count their
    outstanding vector-memory requests by hand, a compiler-inserted scratch load between two LDS-DMA requests would shift the count -- is a
    reload inside the loop.  tools/loop_spills.py reads the device assembly of the two hot units: no scratch instruction in any basic block
    that holds matrix instructions, and no more v_readlane (reloads of spilled scalars) there than the committed baseline."""
    import loop_spills
    hipcc = shutil.which("hipcc") or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    csrc = os.path.join(ROOT, "open-hummingbird-eval_amd", "csrc")
    base = json.load(open(os.path.join(ROOT, "tests", "golden", "loop_spills.json")))
    procs = []
    for unit in ("hbird_knn_bd", "hbird_knn_f16"):
        out = str(tmp_path / f"{unit}.s")
        procs.append((out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-Wno-inline-asm", "-S",
                                             os.path.join(csrc, unit + ".hip"), "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    now = {}
    for out, p in procs:
        assert p.wait() == 0, p.stdout.read().decode()[-2000:]
        now.update(loop_spills.summary(out))
    assert len(now) >= 8, sorted(now)
    problems = []
    for k, d in now.items():
        if d["scratch"]:
            problems.append(f"{k}: {d['scratch']} scratch instructions inside the stage loop")
        if k not in base:
            problems.append(f"{k}: not in tests/golden/loop_spills.json (python tools/loop_spills.py --write-baseline)")
        elif d["readlane"] > base[k]["readlane"]:
            problems.append(f"{k}: {d['readlane']} scalar reloads inside the stage loop > baseline {base[k]['readlane']}")
    assert not problems, "\n".join(problems)
