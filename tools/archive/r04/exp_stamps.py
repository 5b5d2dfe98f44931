#!/usr/bin/env python3
"""Cycle stamps of the fp16 candidate kernel's pool epilogue (an experiment build: lib/abl/libhbird_hip_x_stamps.so, made from a temporary
patch that is not in the tree): where a wave's epilogue cycles go.  usage: exp_stamps.py rows dim nq k"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HBIRD_HIP_LIB"] = os.path.join(ROOT, "open-hummingbird-eval_amd", "lib", "abl", "libhbird_hip_x_stamps.so")
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi import _lib
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq, k = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, D, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
ix.set_fp16(True)
ix.search(q, k)
L = ctypes.CDLL(os.environ["HBIRD_HIP_LIB"])
out = (ctypes.c_uint64 * 16)()
L.hb_x_read_stamps(out, 1)
ix.set_timing(True); ix.search(q, k); ms = ix.last_knn_ms(); ix.set_timing(False)
L.hb_x_read_stamps(out, 0)
v = [int(x) for x in out]
names = ["epilogues", "bulk epilogues", "cycles: scans w/o survivor", "cycles: scans with survivors", "scans with survivors", "survivors (queue)", "cycles: drains", "queue overflows", "cycles: quarter loops", "cycles: compactions", "compactions", "cycles: whole kernel (sum over waves)"]
print(f"{M} x {D}, nq {nq}, k {k}: kernel {ms:.2f} ms")
for n, x in zip(names, v): print(f"  {n:40s} {x:>16,d}")
tot = v[11]
print(f"  share of wave cycles: scans w/o survivor {100 * v[2] / tot:.2f} %, scans with survivors {100 * v[3] / tot:.2f} %, drains {100 * v[6] / tot:.2f} % (of which compactions {100 * v[9] / tot:.2f} %), quarter loops {100 * v[8] / tot:.2f} %")
print(f"  per scan w/o survivor {v[2] / max(1, v[0] - v[1] - v[4]):.0f} cycles; with survivors {v[3] / max(1, v[4]):.0f}; per drain {v[6] / max(1, v[4] - v[7]):.0f}; survivors per drain {v[5] / max(1, v[4]):.2f}; per compaction {v[9] / max(1, v[10]):.0f}")
