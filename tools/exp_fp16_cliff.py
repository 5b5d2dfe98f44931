#!/usr/bin/env python3
"""The use_fp16 cliff on CLUSTERED banks: token worlds -- C class centroids N(0,1)^D, token = centroid[class] + sigma x
N(0,1), bank rows L2-normalised, queries un-normalised tokens of the same world -- at falling sigma: the gap between rank k and rank k' of a
query's scores shrinks against the fp16 rounding bound E, certificates fail, and every failing query used to cost a share of a whole-bank
fp32 search.  Per sigma: share of queries whose first certificate fails, share that reaches the fp32 kernel, ms per search and q-p/s with the
escalation (second fp16 pass, k' = 256, seeded floors) on and off and in the adaptive mode that `use_fp16=True` selects, the plain fp32 search beside them; every result compared bit for bit with
the fp32 search's.
usage: exp_fp16_cliff.py rows dim nq k classes out.json sigma [sigma ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import numpy as np, torch
from hbird_mi.nn.search_hip import HipFlatIndex
M, D, nq, k, C = (int(x) for x in sys.argv[1:6])
out_path = sys.argv[6]
sigmas = [float(x) for x in sys.argv[7:]]
dev = torch.device("cuda", 0)


def timed(ix, q, n=3, warm=2):
    for _ in range(warm):
        ix.search(q, k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(n):
        e0.record(); r = ix.search(q, k); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), r


res = {"rows": M, "dim": D, "queries": nq, "k": k, "classes": C, "world": "token = centroid[class] + sigma x N(0,1); bank rows normalised, queries not", "sigmas": {}}
for sg in sigmas:
    g = torch.Generator(device=dev); g.manual_seed(1234)
    cent = torch.randn((C, D), generator=g, device=dev)
    ix = HipFlatIndex(D, 0, 0); ix.reserve(M); ix.use_current_stream()
    for r in range(0, M, 500_000):
        n = min(500_000, M - r)
        cls = torch.randint(0, C, (n,), generator=g, device=dev)
        ix.add(cent[cls] + sg * torch.randn((n, D), generator=g, device=dev), normalize=True)
    cls = torch.randint(0, C, (nq,), generator=g, device=dev)
    q = cent[cls] + sg * torch.randn((nq, D), generator=g, device=dev)
    t32, (ri, rd) = timed(ix, q, n=2, warm=1)
    row = {"fp32_ms": t32, "fp32_qps": nq / t32 * 1e3}
    for name, mode, on in (("adaptive_mode2", 2, True), ("escalation", 1, True), ("straight_to_fp32", 1, False)):
        ix.set_fp16(mode)                 # 1: the candidate pass always; 2: what use_fp16=True selects -- where it pays, adaptive (hb_launch_knn)
        ix.set_fp16_escalation(on)
        t, (i1, d1) = timed(ix, q, n=5, warm=4)
        row[name] = {"ms": t, "qps": nq / t * 1e3, "first_certificate_failed": ix.last_fp16_escalated(), "reached_fp32": ix.last_fp16_fallbacks(),
                     "share_failed": ix.last_fp16_escalated() / nq, "share_fp32": ix.last_fp16_fallbacks() / nq,
                     "same_bits_as_fp32": bool(torch.equal(i1, ri) and torch.equal(d1.view(torch.int32), rd.view(torch.int32)))}
    res["sigmas"][str(sg)] = row
    print(sg, json.dumps(row), flush=True)
    del ix
    torch.cuda.empty_cache()
json.dump(res, open(out_path, "w"), indent=1)
