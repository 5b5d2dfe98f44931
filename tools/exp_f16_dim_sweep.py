#!/usr/bin/env python3
"""use_fp16 candidate pass against the width D at fixed rows and queries (same tile pairs, same phases): kernel ms = slope x D + constant ->
the stage loop's rate and what a tile pair costs beyond it.  args = queries k rows dim..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nq, k, M = (int(x) for x in sys.argv[1:4])
rows = []
for D in (int(x) for x in sys.argv[4:]):
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ix.set_fp16(True)
    for _ in range(2): ix.search(q, k)
    ms = []
    for _ in range(4):
        ix.set_timing(True); ix.search(q, k); ms.append(ix.last_knn_ms()); ix.set_timing(False)
    info = ix.schedule_info()
    pairs = info["query_tiles"] * info["bank_tiles"] / info["workgroups"]
    ideal = 256 * 256 * D * 2 / (2516.6e12 / 256) * 1e6
    t = float(np.median(ms)) * 1e3 / pairs
    print((M, D, nq, k), f"candidate pass {np.round(ms, 2).tolist()} ms, fallbacks {ix.last_fp16_fallbacks()}; {pairs:.1f} tile pairs per workgroup -> {t:.2f} us per pair ({ideal:.2f} at the nominal fp16 peak = {ideal / t:.3f})", flush=True)
    rows.append((ideal, t))
    del ix
A = np.array([[r[0], 1.0] for r in rows]); b = np.array([r[1] for r in rows])
(sl, c), *_ = np.linalg.lstsq(A, b, rcond=None)
print(f"fit: us per pair = {sl:.3f} x (time at the nominal peak) + {c:.2f} us   (stage loop at {1 / sl:.3f} of the peak; constant per tile pair incl. the phases' boundaries)")
