#!/usr/bin/env python3
"""When and where the workgroups of a kNN launch run (hb_index_wg_stamps): per XCD, how long its workgroups took for the SAME amount of work.
args = rows dim queries k mode[f16|f32] ... (five per case)."""
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
    if os.environ.get("EXP_NO_CLUSTERS"): ix.set_cluster(1, 1, 0)
    for rnd in range(3):
        ix.set_timing(True); ix.search(q, k); kms = ix.last_knn_ms(); st = ix.wg_stamps().astype(np.float64); ix.set_timing(False)
        dur = (st[:, 1] - st[:, 0]) / 100.0      # us
        end = st[:, 1] / 100.0
        print((M, D, nq, k, mode), f"round {rnd}: kernel {kms:.2f} ms; workgroup durations us: min {dur.min():.0f} median {np.median(dur):.0f} max {dur.max():.0f} "
              f"(spread {100 * (dur.max() - dur.min()) / dur.max():.2f} %); starts within {st[:, 0].max() / 100:.1f} us")
        for x in range(8):
            m = st[:, 2] == x
            if m.any():
                print(f"    XCC {x}: {int(m.sum())} blocks, duration median {np.median(dur[m]):.0f} us ({100 * (np.median(dur[m]) / np.median(dur) - 1):+.2f} % of the chip's median), max {dur[m].max():.0f}, last end at {end[m].max():.0f}")
    del ix
