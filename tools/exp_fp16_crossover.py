#!/usr/bin/env python3
"""Whole-search ms (HIP events around hb_index_search, median of 10) with use_fp16 on / off over bank sizes: where does the
certified fp16 mode start to pay?  usage: exp_fp16_crossover.py dim nq k rows..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
D, nq, k = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
for M in (int(x) for x in sys.argv[4:]):
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    out = {}
    for mode in (False, True):
        ix.set_fp16(mode)
        for _ in range(3): ix.search(q, k)
        ms = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ix.search(q, k); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        out[mode] = statistics.median(ms)
    print(f"{M} x {D}, nq {nq}, k {k}: fp32 {out[False]:.2f} ms, use_fp16 {out[True]:.2f} ms", flush=True)
    del ix; torch.cuda.empty_cache()
