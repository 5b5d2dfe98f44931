#!/usr/bin/env python3
"""One-off validation on the GPU box: use_fp16 mode returns the same bits as the fp32 search on the BASELINE shapes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-hummingbird-eval_amd")]
import torch
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda:0")
for (M, D, nq, k, metric) in [(2_074_072, 384, 12_544, 30, 0), (10_000_000, 768, 21_904, 90, 0), (20_345_364, 1024, 21_904, 30, 0),
                              (2_000_000, 768, 21_904, 30, 1)]:
    g = torch.Generator(device=dev).manual_seed(M % 1000)
    ix = HipFlatIndex(D, metric, 0); ix.reserve(M)
    for r in range(0, M, 500_000):
        ix.add(torch.randn((min(500_000, M - r), D), generator=g, device=dev), normalize=True)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    torch.cuda.synchronize(); t = time.time(); i32, d32 = ix.search(q, k); torch.cuda.synchronize(); t32 = time.time() - t
    ix.set_fp16(True); ix.search(q[:256], k)
    torch.cuda.synchronize(); t = time.time(); i16, d16 = ix.search(q, k); torch.cuda.synchronize(); t16 = time.time() - t
    print(f"M={M} D={D} nq={nq} k={k} metric={metric}: equal idx {bool(torch.equal(i32, i16))} dist {bool(torch.equal(d32, d16))} "
          f"fallbacks {ix.last_fp16_fallbacks()}  fp32 {t32*1e3:.0f} ms  fp16 {t16*1e3:.0f} ms  ({t32/t16:.2f}x)", flush=True)
    del ix
