#!/usr/bin/env python3
"""A/B of the L2-sharing cluster schedules on one box: 10 M x 768 bank built once, kNN kernel ms (HIP events) for the
fp32 and the fp16 candidate kernel under each (cluster shape, sync lag), interleaved rounds."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import numpy as np, torch
import bench
from hbird_mi.nn.search_hip import HipFlatIndex

M = int(os.environ.get("EXP_ROWS", 10_000_000)); D = int(os.environ.get("EXP_DIM", 768)); nq = int(os.environ.get("EXP_NQ", 21904)); k = int(os.environ.get("EXP_K", 30))
rounds = int(os.environ.get("EXP_ROUNDS", 2))
cfgs = [tuple(int(x) for x in c.split(",")) for c in os.environ.get("EXP_CFGS", "1,1,0;2,2,0;2,2,6;2,2,12;4,2,6;2,4,6").split(";")]
modes = os.environ.get("EXP_MODES", "f32,f16").split(",")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ref = {}
res = {}
for r in range(rounds + 1):
    for mode in modes:
        ix.set_fp16(mode == "f16")
        for c in cfgs:
            ix.set_cluster(c[0], c[1], c[2])
            ix.set_cluster_sharing(c[3] if len(c) > 3 else 0)       # optional fields: sharing mode, panel tiles, kernel variant
            ix.set_tuning(0, c[4] if len(c) > 4 else 0)
            ix.set_variant(c[5] if len(c) > 5 else 0)
            ix.set_timing(True)
            idx, dist = ix.search(q, k)
            ms = ix.last_knn_ms()
            st = ix.cluster_stats()
            ix.set_timing(False)
            key = (mode,) + c
            if r == 0:
                if mode not in ref: ref[mode] = (idx.clone(), dist.clone())
                ok = torch.equal(idx, ref[mode][0]) and torch.equal(dist, ref[mode][1])
                res[key] = {"same_bits": bool(ok), "ms": [], "schedule": ix.schedule_info()}
            else:
                res[key]["ms"].append(round(ms, 2)); res[key]["sync"] = st
for key, v in res.items():
    print(key, "same_bits", v["same_bits"], "ms", v["ms"], "slots", v["schedule"]["slots"], "panel", v["schedule"].get("panel_tiles"), "cluster", v["schedule"]["cluster"], v.get("sync"), flush=True)
json.dump({str(k_): v for k_, v in res.items()}, open(os.path.join(ROOT, "gpurun_out", os.environ.get("EXP_OUT", "exp_cluster.json")), "w"), indent=1)
