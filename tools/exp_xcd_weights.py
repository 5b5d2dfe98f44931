#!/usr/bin/env python3
"""Per-XCD work shares (hb_index_set_xcd_weights) derived from the workgroups' own durations (hb_index_wg_stamps): kernel ms with equal shares,
then with shares proportional to each XCD group's measured speed, iterated BY HAND (mode 2); the library's own calibration (mode 0) does
the same between searches.  args = rows dim queries k mode[f16|f32] ... (five per case)."""
import os, sys
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
    w = np.ones(8)
    for it in range(5):
        ix.set_xcd_weights(1) if it == 0 else ix.set_xcd_weights(2, w.tolist())
        ms = []
        for _ in range(3):
            ix.set_timing(True); idx, dist = ix.search(q, k); ms.append(ix.last_knn_ms()); st = ix.wg_stamps().astype(np.float64); ix.set_timing(False)
        if ref is None: ref = (idx.clone(), dist.clone())
        assert torch.equal(idx, ref[0]) and torch.equal(dist, ref[1])
        dur = st[:, 1] - st[:, 0]
        grp = np.array([np.median(dur[x::8]) for x in range(8)])          # group x = blocks equal to x mod 8
        xcc = [int(np.bincount(st[x::8, 2].astype(int)).argmax()) for x in range(8)]
        print((M, D, nq, k, mode), f"iteration {it}: shares {np.round(w / w.mean(), 4).tolist()} -> kernel ms {[round(v, 2) for v in ms]}, "
              f"group medians / chip median {np.round(grp / np.median(dur), 4).tolist()} (groups ran on XCC {xcc}), spread {100 * (dur.max() - dur.min()) / dur.max():.2f} %", flush=True)
        # a group that took longer for its share was slower: new share = old share x (median duration of all / its duration)
        w = w * (np.median(grp) / grp)
    del ix
