#!/usr/bin/env python3
"""Is a workgroup's speed a property of its CU?  The headline search several times with calibrated XCD shares: per-workgroup duration relative
to its XCD group's median, correlated between runs.  args = rows dim queries k runs"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
M, D, nq, k, runs = (int(x) for x in sys.argv[1:6])
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
for _ in range(5): ix.search(q, k); torch.cuda.synchronize()
ix.set_xcd_weights(2, ix.xcd_weights()[0])          # freeze the calibrated shares: the same work list in every run
rel = []
for r in range(runs):
    ix.set_timing(True); ix.search(q, k); kms = ix.last_knn_ms(); st = ix.wg_stamps().astype(np.float64); ix.set_timing(False)
    dur = st[:, 1] - st[:, 0]
    grp = np.array([np.median(dur[x::8]) for x in range(8)])
    rel.append(dur / grp[np.arange(len(dur)) % 8] - 1.0)
    print(f"run {r}: kernel {kms:.2f} ms, median workgroup {np.median(dur) / 1e5:.2f} ms, max {dur.max() / 1e5:.2f} ms (+{100 * (dur.max() / np.median(dur) - 1):.3f} %), "
          f"spread inside the XCD groups: std {100 * rel[-1].std():.3f} %, max {100 * rel[-1].max():.3f} %", flush=True)
rel = np.array(rel)
c = np.corrcoef(rel)
print("correlation of the per-workgroup deviations between runs:", np.round(c[np.triu_indices(runs, 1)], 3).tolist())
m = rel.mean(axis=0)
print(f"mean deviation per workgroup: std {100 * m.std():.3f} %, slowest {100 * m.max():.3f} % (block {int(m.argmax())}), fastest {100 * m.min():.3f} %; "
      f"if each workgroup's share followed its mean speed the slowest would end {100 * (rel - m).max(axis=1).mean():.3f} % after its group's median instead of {100 * rel.max(axis=1).mean():.3f} %")
