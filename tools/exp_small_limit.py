#!/usr/bin/env python3
"""Where the small-search list kernel (cold start, register-queue insertions, throttled floor exchange) stops paying: kernel ms with the
built-in limit against other limits (hb_index_set_search_options(ix, 1, stages)).  args = dim queries k limit,limit,... rows..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
D, nq, k = (int(x) for x in sys.argv[1:4]); limits = [int(x) for x in sys.argv[4].split(",")]
for M in (int(x) for x in sys.argv[5:]):
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ref = None; res = {}
    for rnd in range(3):
        for lim in limits:
            ix.set_search_options(True, lim)
            for _ in range(2 if rnd == 0 else 1): idx, dist = ix.search(q, k)
            ix.set_timing(True); idx, dist = ix.search(q, k); res.setdefault(lim, []).append(round(ix.last_knn_ms(), 2)); ix.set_timing(False)
            if ref is None: ref = (idx.clone(), dist.clone())
            assert torch.equal(idx, ref[0]) and torch.equal(dist, ref[1])
    info = ix.schedule_info()
    print((M, D, nq, k), "stages per workgroup", info["query_tiles"] * info["bank_tiles"] // info["workgroups"] * (D // 8), {("built-in" if l == 0 else l): v for l, v in res.items()}, flush=True)
    del ix
