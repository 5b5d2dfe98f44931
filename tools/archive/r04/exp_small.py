#!/usr/bin/env python3
"""Kernel ms of a small search, many repetitions (median / min): rows dim nq k [reps].  HBIRD_HIP_LIB selects the library."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq, k = (int(x) for x in sys.argv[1:5]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
for _ in range(5): ix.search(q, k)
ms = []
for _ in range(reps):
    ix.set_timing(True); ix.search(q, k); ms.append(ix.last_knn_ms()); ix.set_timing(False)
print(os.path.basename(os.environ.get("HBIRD_HIP_LIB", "default")), f"{M} x {D}, nq {nq}, k {k}: median {statistics.median(ms):.3f} ms, min {min(ms):.3f}, max {max(ms):.3f}", flush=True)
