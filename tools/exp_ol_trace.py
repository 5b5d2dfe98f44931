#!/usr/bin/env python3
"""Where a phase boundary of a one-launch search spends its time, per workgroup (hb_index_one_launch_trace: 100 MHz stamps at the arrival at
barrier 1, its pass, the end of the floor computation, the pass of barrier 2).  args = rows dim queries k mode[f16|f32] ... (five per case)."""
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
    ix.set_one_launch(2, inject=3 << 28)
    for _ in range(3):
        ix.set_timing(True); ix.search(q, k); kms = ix.last_knn_ms(); ix.set_timing(False)
    t = ix.one_launch_trace().astype(np.float64) / 100.0          # us
    st = ix.one_launch_stats()
    print((M, D, nq, k, mode), "kernel ms %.3f" % kms, "phases", st["phases"], "given_up", st["given_up"])
    prev_end = None
    for b in range(t.shape[0]):
        arr, p1, fl, p2 = t[b]
        first = arr.min()
        late = np.sort(arr - first)
        line = (f"  boundary {b}: arrivals spread median {np.median(late):6.1f} p90 {late[int(0.9 * len(late))]:6.1f} max {late[-1]:6.1f} us | "
                f"barrier 1 after the last arrival {np.median(p1) - arr.max():5.1f} | floors median {np.median(fl - p1):5.1f} max {(fl - p1).max():5.1f} | "
                f"barrier 2 after the slowest floors {np.median(p2) - fl.max():5.1f} | boundary total (first arrival -> median pass) {np.median(p2) - first:6.1f}")
        if prev_end is not None:
            line += f" | phase before it (median pass -> median arrival) {np.median(arr) - prev_end:8.1f}"
        prev_end = np.median(p2)
        print(line)
        worst = np.argsort(arr)[-5:]
        print("     latest arrivals: blocks", worst.tolist(), "at +", [round(float(arr[w] - first), 1) for w in worst])
    del ix
