#!/usr/bin/env python3
"""use_fp16: what is resident and what the re-rank's row-major copy buys.  Per shape: device memory in use with the fp32
tiles only, after the first use_fp16 search (fp16 tiles, and the row-major fp32 copy where the automatic rule makes it), whole-search ms with
the copy forced on / off where both fit, and the use_fp16 bits against the fp32 search's on a slice of the queries.
usage: exp_fp16_residency.py out.json rows dim nq k [rows dim nq k ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import numpy as np, torch
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0)
out_path = sys.argv[1]
shapes = [tuple(int(x) for x in sys.argv[i:i + 4]) for i in range(2, len(sys.argv), 4)]


def used_gb():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e9


def timed(ix, q, k, n=3, warm=3):
    for _ in range(warm):
        ix.search(q, k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(n):
        e0.record(); r = ix.search(q, k); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), r


res = []
for M, D, nq, k in shapes:
    base = used_gb()
    g = torch.Generator(device=dev); g.manual_seed(5)
    ix = HipFlatIndex(D, 0, 0); ix.reserve(M); ix.use_current_stream()
    for r in range(0, M, 500_000):
        ix.add(torch.randn((min(500_000, M - r), D), generator=g, device=dev), normalize=True)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    row = {"rows": M, "dim": D, "queries": nq, "k": k, "bank_fp32_GB": M * D * 4 / 1e9, "resident_GB_fp32_tiles_only": used_gb() - base}
    sub = q[:2048].contiguous()
    ri, rd = ix.search(sub, k)                                   # the fp32 kernel on a slice: the bits to match
    ix.set_fp16(2)
    for name, mode in (("automatic", 0), ("copy_never", 2), ("copy_always", 1)):
        if mode == 1 and M * D * 4 * 2.6 > 0.9 * torch.cuda.mem_get_info(0)[1]:
            row[name] = "does not fit"
            continue
        ix.set_rerank_copy(mode)
        t, (i1, d1) = timed(ix, q, k)
        row[name] = {"ms": t, "qps": nq / t * 1e3, "resident_GB": used_gb() - base, "rerank_copy_GB": ix.rerank_copy_bytes() / 1e9,
                     "first_certificate_failed": ix.last_fp16_escalated(), "reached_fp32": ix.last_fp16_fallbacks(),
                     "same_bits_as_fp32_on_2048_queries": bool(torch.equal(i1[:2048], ri) and torch.equal(d1[:2048].view(torch.int32), rd.view(torch.int32)))}
    ix.set_rerank_copy(0)
    res.append(row)
    print(json.dumps(row), flush=True)
    del ix, q
    torch.cuda.empty_cache()
json.dump(res, open(out_path, "w"), indent=1)
