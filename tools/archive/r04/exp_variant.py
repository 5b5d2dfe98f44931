#!/usr/bin/env python3
"""A/B of kNN kernel variants (hb_index_set_variant) on one bank, interleaved rounds: rows dim nq k "v1,v2,..." [fp16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq, k = (int(x) for x in sys.argv[1:5])
variants = [int(x) for x in sys.argv[5].split(",")]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ix.set_fp16(len(sys.argv) > 6)
res = {v: [] for v in variants}; ref = None; same = {}
for r in range(4):
    for v in variants:
        ix.set_variant(v); ix.set_timing(True)
        i, d = ix.search(q, k); ms = ix.last_knn_ms(); ix.set_timing(False)
        if ref is None: ref = (i.clone(), d.clone())
        same[v] = bool(torch.equal(i, ref[0]) and torch.equal(d, ref[1]))
        if r: res[v].append(round(ms, 1))
for v in variants: print("variant", v, res[v], "same bits", same[v], flush=True)
