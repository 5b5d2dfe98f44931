#!/usr/bin/env python3
"""The library's own per-XCD calibration (hb_index_set_xcd_weights mode 0) search after search, beside equal shares (mode 1), for either
kernel family: kernel ms per search, the shares in use, and that the results never change.  args = rows dim queries k mode[f16|f32] ...
(Both kernel families calibrate shares of their own; profiles/r05/xcd_auto_fp16_negative.txt is this script on an earlier build whose phased lists
had common cuts: there the fp16 kernel got slower with shares.)"""
import os, sys, time
import numpy as np
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
    ref = None
    for label, xmode, reps in (("equal", 1, 4), ("calibrated", 0, 8), ("equal", 1, 3), ("calibrated", 0, 4)):
        ix.set_xcd_weights(xmode)
        for r in range(reps):
            ix.set_timing(True); t0 = time.perf_counter(); idx, dist = ix.search(q, k); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
            ms = ix.last_knn_ms(); ix.set_timing(False)
            if ref is None: ref = (idx.clone(), dist.clone())
            assert torch.equal(idx, ref[0]) and torch.equal(dist, ref[1])
            w, rounds = ix.xcd_weights(mode == "f16")
            print((M, D, nq, k, mode), label, f"search {r}: kernel {ms:.2f} ms, whole search {wall:.2f} ms, shares {np.round(w, 4).tolist()} after {rounds} rounds", flush=True)
    del ix
