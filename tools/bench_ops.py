#!/usr/bin/env python3
"""Time the secondary hot-path kernels (K1, K2, K3, K5, K6, K7) at cfg-2 / cfg-3 shapes and report achieved GB/s
against their algorithmic bytes.  GPU only."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-hummingbird-eval_amd")]
import torch
from hbird_mi import ops
from hbird_mi.nn.search_hip import HipFlatIndex

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = []
for name, (B, S, ps, D, C, k) in {"cfg2": (64, 14, 16, 384, 21, 30), "cfg3": (16, 37, 14, 768, 151, 30)}.items():
    N, H = S * S, S * ps
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.randn((B * N, D), generator=g, device=dev)
    y = torch.randint(0, C, (B, 1, H, H), generator=g, device=dev)
    # K1: fused normalise + append (8*D bytes per row: read once + write once)
    ix = HipFlatIndex(D, 0, 0); ix.reserve(B * N * 8); ix.use_current_stream()
    def k1():
        ix.reset() if ix.ntotal + B * N > B * N * 8 else None
        ix.add(feats, normalize=True)
    ms = timeit(k1); res.append((name, "K1 normalize+append", ms, 8 * D * B * N))
    # K2
    ms = timeit(lambda: ops.patch_label_hist(y, ps, C, map255=True)); res.append((name, "K2 patch_label_hist", ms, (8 * ps * ps + 4 * C) * B * N))
    lab = ops.patch_label_hist(y, ps, C).view(B, N, C)
    # K3
    sc, ne, nz = ops.patch_scores(lab)
    r = torch.rand(B * N, device=dev); roff = (torch.arange(B, device=dev) * N)
    ms = timeit(lambda: ops.patch_scores(lab)); res.append((name, "K3a patch_scores", ms, 4 * C * B * N))
    ms = timeit(lambda: ops.patch_select(sc, ne, r, roff, min(N, 100))); res.append((name, "K3b patch_select", ms, 8 * B * N))
    # K5 on a 1M-row bank
    M = 1_000_000
    bank = HipFlatIndex(D, 0, 0); bank.reserve(M); bank.use_current_stream()
    cbank = HipFlatIndex(8, 0, 0); cbank.use_current_stream(); cbank.set_label_denominator(ps * ps); cbank.set_num_classes(C)   # the same labels as uint16 counts
    for r0 in range(0, M, 250_000):
        bank.add(torch.randn((250_000, D), generator=g, device=dev), normalize=True)
        labs = torch.randint(0, ps * ps + 1, (250_000, C), generator=g, device=dev).float() / torch.tensor(float(ps * ps), device=dev)   # values j / P, like K2's
        bank.add_labels(labs); cbank.add_labels(labs)
    bank.set_num_classes(C)
    q = 3 * torch.randn((B * N, D), generator=g, device=dev)
    idx, dist = bank.search(q, k)
    ms = timeit(lambda: bank.aggregate(q, idx, dist)); res.append((name, "K5 aggregate (+query norms), fp32 labels", ms, (4 * k * C + 12 * k + 4 * C + 4 * D) * B * N))
    lh = bank.aggregate(q, idx, dist).view(B, N, C)
    agg = HipFlatIndex(D, 0, 0); agg.use_current_stream(); agg.set_label_count_table(cbank.copy_label_counts(), bank.copy_norms(), ps * ps, 0)
    ms = timeit(lambda: agg.aggregate(q, idx, dist)); res.append((name, "K5 aggregate (+query norms), uint16 counts, borrowed dense table", ms, (2 * k * C + 12 * k + 4 * C + 4 * D) * B * N))
    assert torch.equal(agg.aggregate(q, idx, dist).view(torch.int32), lh.reshape(-1, C).view(torch.int32))
    del agg
    # the index's OWN count table (rows padded to 16 bytes: the wide gather) -- what a single-GPU evaluation runs
    own = HipFlatIndex(D, 0, 0); own.reserve(M); own.use_current_stream(); own.set_label_denominator(ps * ps)
    g2 = torch.Generator(device=dev).manual_seed(0)
    torch.randn((B * N, D), generator=g2, device=dev); torch.randint(0, C, (B, 1, H, H), generator=g2, device=dev)      # replay the stream up to the bank rows
    for r0 in range(0, M, 250_000):
        own.add(torch.randn((250_000, D), generator=g2, device=dev), normalize=True)
        own.add_labels(torch.randint(0, ps * ps + 1, (250_000, C), generator=g2, device=dev).float() / torch.tensor(float(ps * ps), device=dev))
    own.set_num_classes(C)
    ms = timeit(lambda: own.aggregate(q, idx, dist)); res.append((name, "K5 aggregate (+query norms), uint16 counts, own padded table", ms, (2 * k * C + 12 * k + 4 * C + 4 * D) * B * N))
    assert torch.equal(own.aggregate(q, idx, dist).view(torch.int32), lh.reshape(-1, C).view(torch.int32))
    del own, cbank
    # K6
    ms = timeit(lambda: ops.upsample_argmax(lh, S, H, H)); res.append((name, "K6 upsample_argmax", ms, 4 * C * N * B + 8 * H * H * B))
    pred = ops.upsample_argmax(lh, S, H, H)
    conf = torch.zeros((C, C), dtype=torch.int64, device=dev)
    ms = timeit(lambda: ops.confusion_update(conf, y, pred, 255)); res.append((name, "K7 confusion, noise masks", ms, 16 * H * H * B))
    ms = timeit(lambda: ops.upsample_argmax_confusion(lh, S, y, conf, 255)); res.append((name, "K6 + K7 fused (no class map), noise masks", ms, 4 * C * N * B + 8 * H * H * B))
    # piecewise-constant masks and predictions (what segmentation data looks like): rectangles of one class, label_hat peaked per patch
    import sys as _s; _s.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_inputs as gi
    yr = torch.from_numpy(gi.random_masks(B, H, H, C, seed=3, with_255=True)).to(dev)
    lhr = ops.patch_label_hist(yr, ps, C, map255=True).view(B, N, C)
    predr = ops.upsample_argmax(lhr, S, H, H)
    ms = timeit(lambda: ops.confusion_update(conf, yr, predr, 255)); res.append((name, "K7 confusion, rectangle masks", ms, 16 * H * H * B))
    ms = timeit(lambda: ops.upsample_argmax_confusion(lhr, S, yr, conf, 255)); res.append((name, "K6 + K7 fused (no class map), rectangle masks", ms, 4 * C * N * B + 8 * H * H * B))
    ms = timeit(lambda: ops.upsample_argmax(lhr, S, H, H)); res.append((name, "K6 upsample_argmax, rectangle masks", ms, 4 * C * N * B + 8 * H * H * B))
    del bank, ix
rows = []
for name, op, ms, nbytes in res:
    print(f"{name}  {op:30s} {ms:8.3f} ms  {nbytes / 1e6:9.1f} MB algorithmic  {nbytes / ms / 1e6:8.1f} GB/s  "
          f"{nbytes / ms / 1e6 / 8000:6.3f} of 8 TB/s")
    rows.append({"shape": name, "kernel": op, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 2),
                 "GBps": round(nbytes / ms / 1e6, 1), "frac_of_hbm_8TBps": round(nbytes / ms / 1e6 / 8000, 4)})
if len(sys.argv) > 1:      # tools/bench_ops.py <out.json>: the secondary-kernel table of profiles/<round>/README.md (kept under profiles/<round>/)
    json.dump({"device": torch.cuda.get_device_name(0), "timing": "torch.cuda.Event over 5 calls after one warm-up",
               "rows": rows}, open(sys.argv[1], "w"), indent=1)
