#!/usr/bin/env python3
"""One search per cluster configuration (argv: rows fp16 "cq,cb,lag;..."), for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M = int(sys.argv[1]); fp16 = int(sys.argv[2]); cfgs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[3].split(";")]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(768, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, 768, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((21904, 768), generator=g, device=dev)
ix.set_fp16(bool(fp16))
for c in cfgs:
    ix.set_cluster(*c[:3])
    ix.set_cluster_sharing(c[3] if len(c) > 3 else 0)
    ix.set_tuning(0, c[4] if len(c) > 4 else 0)
    ix.set_variant(c[5] if len(c) > 5 else 0)
    ix.search(q, 30); torch.cuda.synchronize()
