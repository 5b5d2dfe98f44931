#!/usr/bin/env python3
"""A sweep over search shapes to find performance cliffs: for every bank (rows x dim) built once, searches with nq in {12544, 21904},
k in {30, 90}, fp32 and use_fp16: kernel ms (HIP events inside the library), whole-search ms (HIP events around 2 searches), the kernel's
fraction of its MFMA peak (157.3 TFLOP/s fp32, 2516.6 fp16) and the share of the search spent outside the kNN kernel.
usage: sweep_shapes.py out.json [rows,dim ...]   (SWEEP_NQ="196,1369": other query counts)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
banks = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [
    (50_176, 384), (300_000, 384), (300_000, 768), (1_250_000, 768), (2_074_072, 384), (2_500_000, 1024), (5_000_000, 384), (5_000_000, 768), (10_000_000, 768)]
out = []
for M, D in banks:
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    for nq in [int(x) for x in os.environ.get("SWEEP_NQ", "12544,21904").split(",")]:
        g = torch.Generator(device=dev); g.manual_seed(7)
        q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
        for k in (30, 90):
            for mode in ("f32", "f16"):
                ix.set_fp16(mode == "f16")
                ix.search(q, k)                                  # warm-up (fp16 copy, schedule)
                ix.set_timing(True); ix.search(q, k); kms = ix.last_knn_ms(); ix.set_timing(False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ix.search(q, k); ix.search(q, k); e1.record(); torch.cuda.synchronize()
                sms = e0.elapsed_time(e1) / 2
                peak = 157.3 if mode == "f32" else 2516.6
                frac = 2.0 * M * nq * D / (kms * 1e-3) / 1e12 / peak
                info = ix.schedule_info()
                rec = dict(rows=M, dim=D, nq=nq, k=k, mode=mode, kernel_ms=round(kms, 3), search_ms=round(sms, 3), frac=round(frac, 3),
                           outside=round(1 - kms / sms, 3), cluster=info["cluster"], slots=info["slots"], fallbacks=ix.last_fp16_fallbacks() if mode == "f16" else 0)
                out.append(rec)
                print(f"{M:>9} x {D:<4} nq {nq:<5} k {k:<2} {mode}: kernel {kms:9.2f} ms  frac {frac:5.3f}  search {sms:9.2f} ms  outside {100 * (1 - kms / sms):5.1f} %  cluster {info['cluster']} slots {info['slots']}", flush=True)
    del ix
    torch.cuda.empty_cache()
json.dump(out, open(sys.argv[1], "w"), indent=1)
