#!/usr/bin/env python3
"""Stress of the small-search LIST kernel (120 k - 400 k stages per workgroup: the shapes the fuzz does not reach): the same search many
times, other workgroup counts and XCD shares in between -- every run must return the first run's bits, and those must be the chain oracle's on
a query sample.  args = rows dim queries k repeats"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from hbird_mi.nn.search_hip import HipFlatIndex
from helpers import chain_oracle_topk_chunked
M, D, nq, k, reps = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = torch.Generator(device=dev).manual_seed(M)
ix = HipFlatIndex(D, 0, 0); ix.reserve(M)
for r in range(0, M, 500_000):
    n = min(500_000, M - r); ix.add(torch.randn((n, D), generator=g, device=dev), normalize=True)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
idx0, dist0 = ix.search(q, k)
sel = torch.linspace(0, nq - 1, 256, device=dev).long()
ci, cd = chain_oracle_topk_chunked(ix, q[sel], M, k)
ok = np.array_equal(idx0[sel].cpu().numpy(), ci) and np.array_equal(dist0[sel].cpu().numpy().view(np.uint32), cd.view(np.uint32))
print((M, D, nq, k), "first run equals the chain oracle on 256 queries:", ok, ix.schedule_info(), flush=True)
bad = 0
rng = np.random.default_rng(1)
for r in range(reps):
    if r % 5 == 4: ix.set_tuning(int(rng.choice([0, 256, 248, 192, 128])), 0)
    if r % 7 == 6: ix.set_xcd_weights(2, rng.uniform(0.9, 1.1, size=8).tolist())
    idx, dist = ix.search(q, k)
    same = torch.equal(idx, idx0) and torch.equal(dist, dist0)
    if not same:
        bad += 1
        d = (idx != idx0).any(dim=1).nonzero().flatten()
        print(f"run {r}: {d.numel()} queries differ, first {d[:5].tolist()}", flush=True)
print("runs", reps, "different from the first:", bad)
sys.exit(1 if bad or not ok else 0)
