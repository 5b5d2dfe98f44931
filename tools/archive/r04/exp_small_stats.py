#!/usr/bin/env python3
"""Diagnostic (-DKN_STAMPS build): max / mean per-workgroup cycles of the small-search kernel through the stats words."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
from hbird_mi import _lib
M, D, nq, k = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
for r in range(3):
    ix.set_timing(True); ix.search(q, k); ms = ix.last_knn_ms(); ix.set_timing(False)
    out = (ctypes.c_int64 * 4)(); _lib.check(_lib.lib().hb_index_cluster_stats(ix._h, out))
    G = ix.schedule_info()["workgroups"]
    print(f"kernel {ms:.2f} ms; per workgroup: max {out[0] * 1024 / 1e6:.2f} M cycles, mean {out[1] * 1024 / G / 1e6:.2f} M, epilogue mean {out[2] * 1024 / G / 1e6:.2f} M, candidates per wave-0 mean {out[3] / G:.0f}", ix.schedule_info(), flush=True)
