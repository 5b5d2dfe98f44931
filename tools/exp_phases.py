#!/usr/bin/env python3
"""Kernel ms (HIP events, best of 3 after a warm-up) of pool searches: args = rows dim queries k mode[f16|f32] ... (five per case).
EXP_PHASES=0 runs them unphased, EXP_VARIANT=<n> selects a kernel variant (hb_index_set_search_options / hb_index_set_variant)."""
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
    ix.set_search_options(phases=os.environ.get("EXP_PHASES", os.environ.get("HBIRD_PHASES", "1")) != "0")
    ix.set_variant(int(os.environ.get("EXP_VARIANT", os.environ.get("HBIRD_KNN_VARIANT", "0"))))
    ms = []
    for r in range(4):
        ix.set_timing(True); idx, dist = ix.search(q, k); ms.append(round(ix.last_knn_ms(), 2)); ix.set_timing(False)
    print("phases", os.environ.get("EXP_PHASES", os.environ.get("HBIRD_PHASES", "1")), (M, D, nq, k, mode), "ms", ms, "checksum", int(idx.sum()), float(dist.double().sum()), flush=True)
    del ix
