#!/usr/bin/env python3
"""One launch per phased search vs a launch per phase: kernel ms (HIP events around the kNN launches) and whole-search ms (HIP events
around hb_index_search, 20 searches) for a list of searches, interleaved, with the barrier statistics of block 0.
args = rows dim queries k mode[f16|f32] ... (five per case)."""
import os, sys
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
    res = {}
    for rnd in range(3):
        for name, m in (("per_phase", 1), ("one_launch", 2)):
            ix.set_one_launch(m)
            ix.set_timing(True); idx, dist = ix.search(q, k); kms = ix.last_knn_ms(); ix.set_timing(False)
            st = ix.one_launch_stats()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                idx, dist = ix.search(q, k)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append((round(kms, 3), round(e0.elapsed_time(e1) / 20, 3)))
            res[name + "_stats"] = st
            res[name + "_sum"] = (int(idx.sum()), float(dist.double().sum()))
    st = res["one_launch_stats"]
    nb = max(1, st["boundaries"])
    print((M, D, nq, k, mode), "per-phase launches (kernel ms, search ms):", res["per_phase"], "| one launch:", res["one_launch"],
          "| phases", st["phases"], "per boundary us: barrier1 %.1f floors %.1f barrier2 %.1f" % (st["barrier1_ticks"] / nb / 100, st["floor_ticks"] / nb / 100, st["barrier2_ticks"] / nb / 100),
          "given_up", st["given_up"], "same bits" if res["per_phase_sum"] == res["one_launch_sum"] else "DIFFERENT RESULTS", flush=True)
    del ix
