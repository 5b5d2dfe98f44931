#!/usr/bin/env python3
"""Time the fp16 candidate kernel (HIP events) for the library named by HBIRD_HIP_LIB: rows dim variant..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ix.set_fp16(True)
out = []
cl = [int(x) for x in os.environ.get("EXP_CL", "1,1,0").split(",")]
ix.set_cluster(*cl)
for v in [int(x) for x in sys.argv[4].split(",")]:
    ix.set_variant(v)
    ms = []
    for r in range(3):
        ix.set_timing(True); ix.search(q, 30); ms.append(round(ix.last_knn_ms(), 1)); ix.set_timing(False)
    out.append((v, ms))
print(os.path.basename(os.environ.get("HBIRD_HIP_LIB", "default")), "cluster", cl, out, ix.cluster_stats(), flush=True)
