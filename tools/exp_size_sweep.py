#!/usr/bin/env python3
"""fp32 kernel time against bank rows at a fixed query batch (no clusters, calibrated XCD shares): is there a per-search constant?
Prints kernel ms, the workgroups' median / max duration and us per (256 x 256) tile pair.  args = dim queries k rows..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
D, nq, k = (int(x) for x in sys.argv[1:4])
for M in (int(x) for x in sys.argv[4:]):
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ix.set_cluster(1, 1, 0)
    for _ in range(4): ix.search(q, k); torch.cuda.synchronize()
    info = ix.schedule_info()
    tiles = info["query_tiles"] * info["bank_tiles"] / info["workgroups"]
    for rnd in range(3):
        ix.set_timing(True); ix.search(q, k); kms = ix.last_knn_ms(); st = ix.wg_stamps().astype(np.float64); ix.set_timing(False)
        dur = (st[:, 1] - st[:, 0]) / 100.0
        ideal = 256 * 256 * D * 2 / (157.3e12 / 256) * 1e6
        print((M, D, nq, k), f"kernel {kms:.2f} ms; workgroups: first start -> last end {(st[:, 1].max() - st[:, 0].min()) / 1e5:.2f} ms, duration median {np.median(dur) / 1e3:.2f} max {dur.max() / 1e3:.2f} ms; "
              f"{tiles:.1f} tile pairs per workgroup -> {np.median(dur) / tiles:.2f} us per pair (median workgroup; {ideal:.2f} at the nominal peak = {ideal * tiles / np.median(dur):.4f}); "
              f"shares {np.round(ix.xcd_weights()[0], 4).tolist()}", flush=True)
    # back to back: two searches enqueued without a host sync between them
    ix.set_timing(True); ix.search(q, k); ix.search(q, k); kms2 = ix.last_knn_ms(); ix.set_timing(False)
    print((M, D, nq, k), f"second of two searches enqueued back to back: kernel {kms2:.2f} ms", flush=True)
    del ix
