#!/usr/bin/env python3
"""One-off differential fuzz of the small / medium search paths against the CPU oracle (bit-exact indices and distances):
random shapes, both metrics, odd workgroup counts, lists and pools, fp16 mode on and off.  usage: python tests/fuzz_small.py [cases] [seed]   (lives under tests/: it uses the oracle as the checker)
FUZZ_MID=1: few workgroups (8 / 16) on 300 k - 700 k rows at D = 384 / 768, k <= 32: 120 k - 400 k stages per workgroup, the small-search LIST kernel with its
quota-floor exchange (the default cases never reach it: they run on pools).
FUZZ_TIGHT=1: token-world banks (class centroids + sigma x noise, sigma in {0.3, 0.1, 0.03}) searched with use_fp16 in mode 1 or 2, the escalation on or
off, three searches per index: certificates fail, the second fp16 pass, the fp32 fallback and the adaptive paths (wide-first, fp32 right away) all run.
FUZZ_BIGK=1: k in {257, 300, 512, 600, 1100}: passes behind a ceiling (hb_launch_knn_bigk), also with use_fp16 set.
FUZZ_XCD=1: random per-XCD work shares in [0.8, 1.25] (hb_index_set_xcd_weights(ix, 2, w8): weighted work lists), fp32 and use_fp16 searches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import numpy as np, torch
import oracle
from hbird_mi.nn.search_hip import HipFlatIndex
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(n_cases):
    D = int(rng.choice([32, 64, 96, 100, 384]))
    M = int(rng.integers(300, int(os.environ.get("FUZZ_MAX_ROWS", 150_000)) if D < 384 else int(os.environ.get("FUZZ_MAX_ROWS", 150_000)) * 2 // 5))
    nq = int(rng.integers(1, 3000))
    k = int(rng.choice([1, 5, 30, 32, 40, 90]))
    metric = int(rng.integers(0, 2))
    if os.environ.get("FUZZ_MID"):
        D = int(rng.choice([384, 768])); M = int(rng.integers(300_000, 700_000)); nq = int(rng.integers(1000, 2300)); k = int(rng.choice([1, 5, 8, 30, 32]))
    if os.environ.get("FUZZ_BIGK"):
        k = int(rng.choice([257, 300, 512, 600, 1100])); nq = int(rng.integers(1, 600))
    G = int(rng.choice([0, 0, 17, 64, 256])) if not os.environ.get("FUZZ_MID") else int(rng.choice([8, 16]))
    fp16 = bool(rng.integers(0, 2)) and k <= 128 and not os.environ.get("FUZZ_MID")
    bank = rng.standard_normal((M, D), dtype=np.float32); bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    if rng.integers(0, 2): bank[rng.integers(0, M, size=50)] = bank[0]          # duplicates: ties by id
    q = (3.0 * rng.standard_normal((nq, D))).astype(np.float32)
    tight = bool(os.environ.get("FUZZ_TIGHT"))
    if tight:
        C = int(rng.choice([3, 21])); sg = float(rng.choice([0.3, 0.1, 0.03]))
        cent = rng.standard_normal((C, D)).astype(np.float32)
        bank = cent[rng.integers(0, C, size=M)] + sg * rng.standard_normal((M, D), dtype=np.float32); bank /= np.linalg.norm(bank, axis=1, keepdims=True)
        q = (cent[rng.integers(0, C, size=nq)] + sg * rng.standard_normal((nq, D))).astype(np.float32)
        k = int(rng.choice([5, 30, 64, 90])); fp16 = True
    if rng.integers(0, 3) == 0:                                                   # non-finite values (tests/test_edge_gpu.py): NaN rows, NaN / inf queries
        bank[rng.integers(0, M, size=max(1, M // 50))] = np.nan
        q[rng.integers(0, nq, size=max(1, nq // 100)), rng.integers(0, D)] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
    variant = int(rng.choice([0, 0, 3, 4, 6])); cl = [(0, 0, -1), (0, 0, -1), (2, 2, 4), (2, 4, 16)][int(rng.integers(0, 4))]
    ix = HipFlatIndex(D, metric, 0); ix.add(torch.from_numpy(bank).cuda()); ix.set_fp16((int(rng.integers(1, 3)) if fp16 else 0) if tight or os.environ.get("FUZZ_BIGK") else fp16); ix.set_tuning(G, 0)
    if tight: ix.set_fp16_escalation(bool(rng.integers(0, 4)))
    ix.set_variant(variant); ix.set_cluster(*cl)
    if os.environ.get("FUZZ_XCD"): ix.set_xcd_weights(2, rng.uniform(0.8, 1.25, size=8).tolist())
    for _ in range(3 if tight else 1):          # (mode 2 adapts between searches: every path must give the same bits)
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        if tight:
            i0 = idx if _ == 0 else i0; d0 = dist if _ == 0 else d0
            assert torch.equal(idx, i0) and torch.equal(dist.view(torch.int32), d0.view(torch.int32)), "searches of one index differ"
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, "dot_product" if metric == 0 else "l2", 0)
    ok = np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dist.cpu().numpy().view(np.uint32), rdist.view(np.uint32))
    bad += not ok
    print(f"case {c}: M {M} D {D} nq {nq} k {k} metric {metric} G {G} fp16 {fp16} variant {variant} cluster {cl} -> {'ok' if ok else 'MISMATCH'} {ix.schedule_info()['slots']} slots, {(lambda i: i['query_tiles'] * i['bank_tiles'] // max(1, i['workgroups']) * ((D + 7) // 8))(ix.schedule_info())} stages per workgroup" + (f", sigma {sg} failed first {ix.last_fp16_escalated()} reached fp32 {ix.last_fp16_fallbacks()}" if tight else ""), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
