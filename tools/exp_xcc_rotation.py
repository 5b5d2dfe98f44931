#!/usr/bin/env python3
"""Which XCD does block 0 of a kNN launch land on?  (HIP promises nothing; observed: round-robin over the XCDs -- from where?)  Searches with
small torch kernels of 1 .. 7 workgroups in between; prints the XCC ids of blocks 0-15 of every search's last launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import numpy as np, torch
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0)
M, D, nq, k = 200_000, 384, 12_544, 30
g = torch.Generator(device=dev); g.manual_seed(1)
ix = HipFlatIndex(D, 0, 0); ix.add(torch.randn((M, D), generator=g, device=dev), normalize=True)
q = torch.randn((nq, D), generator=g, device=dev)
ix.set_timing(True)
x = torch.zeros(64 * 7, device=dev)
for i in range(24):
    nb = i % 8
    if nb:
        y = x[: 64 * nb] + 1.0          # one small elementwise kernel (its grid: whatever torch picks for 64 * nb elements)
    ix.search(q, k); torch.cuda.synchronize()
    t = ix.wg_stamps()
    print(f"search {i:2d} after a kernel on {64 * nb:3d} elements: XCC of blocks 0-15 {t[:16, 2].tolist()}  groups consistent {bool((t[:, 2] == t[np.arange(len(t)) % 8, 2]).all())} distinct {len(set(t[:8, 2].tolist()))}", flush=True)
