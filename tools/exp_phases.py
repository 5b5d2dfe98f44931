#!/usr/bin/env python3
"""Kernel ms (HIP events, best of 3 after a warm-up) of pool searches for the HBIRD_PHASES setting of the environment:
args = rows dim queries k mode[f16|f32] ... (five per case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
a = sys.argv[1:]
for i in range(0, len(a), 5):
    M, D, nq, k = (int(x) for x in a[i:i + 4]); mode = a[i + 4]
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ix.set_fp16(mode == "f16")
    ms = []
    for r in range(4):
        ix.set_timing(True); idx, dist = ix.search(q, k); ms.append(round(ix.last_knn_ms(), 2)); ix.set_timing(False)
    print("phases", os.environ.get("HBIRD_PHASES", "1"), (M, D, nq, k, mode), "ms", ms, "checksum", int(idx.sum()), float(dist.double().sum()), flush=True)
    del ix
