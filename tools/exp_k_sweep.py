#!/usr/bin/env python3
"""Kernel ms of use_fp16 / fp32 searches against k (one bank): where the candidate volume (k' = 2k rounded up to 64, pools of max(2k', k' + 128))
and the pool instantiation (<4> up to 256 entries, <8> beyond) cost.  usage: exp_k_sweep.py rows dim nq "k1,k2,..." [f16|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq = (int(x) for x in sys.argv[1:4]); ks = [int(x) for x in sys.argv[4].split(",")]; mode = sys.argv[5] if len(sys.argv) > 5 else "f16"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ix.set_fp16(mode == "f16")
for k in ks:
    ix.search(q, k)
    ms = []
    for _ in range(3):
        ix.set_timing(True); ix.search(q, k); ms.append(round(ix.last_knn_ms(), 2)); ix.set_timing(False)
    print(f"{M} x {D}, nq {nq}, {mode}, k {k:3d}: kernel ms {ms}", flush=True)
