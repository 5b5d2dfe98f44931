#!/usr/bin/env python3
"""Un-profiled kernel clock, share calibration and its guard over a run of searches (hb_index_kernel_clock, hb_index_xcd_stats).
usage: exp_clock_guard.py rows dim nq k searches [fp16]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq, k, n = (int(x) for x in sys.argv[1:6])
fp16 = len(sys.argv) > 6 and sys.argv[6] == "fp16"
dev = torch.device("cuda", 0)
ix = HipFlatIndex(D, 0, 0); ix.reserve(M)
g = torch.Generator(device=dev); g.manual_seed(1)
for r in range(0, M, 500_000):
    ix.add(torch.randn((min(500_000, M - r), D), generator=g, device=dev), normalize=True)
ix.set_fp16(fp16)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ix.set_timing(True)
for i in range(n):
    ix.search(q, k); torch.cuda.synchronize()
    if os.environ.get("DUMP_STAMPS"):
        import numpy as np
        t = ix.wg_stamps()
        d = (t[:, 1] - t[:, 0]) & 0xFFFFFFFF
        print("per-XCD durations (us) min/median/max:", [(round(float(d[x::8].min()) / 100, 1), round(float(np.median(d[x::8])) / 100, 1), round(float(d[x::8].max()) / 100, 1)) for x in range(8)],
              "xcc ok", bool((t[:, 2] == np.arange(len(t)) % 8).all()), flush=True)
    print(json.dumps({"search": i, "knn_ms": round(ix.last_knn_ms(), 3), "clock": {a: round(b, 4) for a, b in ix.kernel_clock().items()},
                      "cluster": ix.schedule_info()["cluster"], "shares": [round(v, 4) for v in ix.xcd_weights(fp16)[0]], "stats": ix.xcd_stats(fp16)}), flush=True)
