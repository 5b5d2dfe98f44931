#!/usr/bin/env python3
"""Parity at scale ("Parity sets"), run on the GPU box:

a token-level segmentation world -- C class centroids N(0,1)^D, a patch token = its class histogram x centroids +
0.5 N(0,1), masks = random rectangles -- gives a cfg-2-sized bank (10,582 train images x 196 patches = 2,074,072 rows,
D = 384, C = 21) and 64 validation images (12,544 queries).  The engine (HbirdEvaluation over the C ABI) is compared
with the CPU oracle on the SAME bank:
  * kNN indices vs the float64 definition (ordered / set / near-tie-excused match rates, 4 ulp) and vs the fp32 chain
    oracle (bit-exact),
  * label_hat vs the oracle's cross-attention (max abs difference),
  * mIoU vs the oracle's metric.
Writes one JSON (default gpurun_out/parity_at_scale.json).  Test infrastructure (it lives under tests/ because it uses
the oracle, as the checker): run as `python tests/parity_report.py`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-hummingbird-eval_amd"), os.path.join(ROOT, "tests")]
import numpy as np   # noqa: E402
import torch         # noqa: E402
import oracle        # noqa: E402
import golden_inputs as gi   # noqa: E402
from hbird_mi.hbird_eval import HbirdEvaluation   # noqa: E402


class TokenWorld(torch.nn.Module):
    """Fake extractor: the tokens of a batch are a function of its masks (sent through channel 0 of x)."""

    def __init__(self, C, D, H, ps, seed):
        super().__init__()
        self.C, self.D, self.H, self.ps = C, D, H, ps
        self.eval_spatial_resolution = H // ps
        self.d_model = D
        rng = np.random.default_rng(seed)
        self.centroids = torch.from_numpy(rng.standard_normal((C, D)).astype(np.float32))
        self.gen = torch.Generator().manual_seed(seed + 1)
        self.log = None

    def forward_features(self, x):
        y = torch.round(x[:, 0] * 255.0).long().cpu()                  # masks ride in channel 0 as mask / 255
        y[y == 255] = 0
        B = y.shape[0]
        hist = oracle.patch_label_hist(y[:, None].numpy(), self.ps, self.C).reshape(B, -1, self.C)
        tok = torch.from_numpy(hist) @ self.centroids
        tok = tok + 0.5 * torch.randn(tok.shape, generator=self.gen)
        if self.log is not None:
            self.log.append(tok.numpy().copy())
        return tok.to(x.device), None


def loader(n_images, B, H, C, seed, with_255):
    out = []
    for i in range(0, n_images, B):
        b = min(B, n_images - i)
        y = gi.random_masks(b, H, H, C, seed + i, with_255=with_255).astype(np.float32) / np.float32(255.0)
        x = torch.zeros((b, 3, H, H))
        yt = torch.from_numpy(y)
        x[:, 0] = yt[:, 0]
        out.append((x, yt))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-images", type=int, default=10_582)
    ap.add_argument("--val-images", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_at_scale.json"))
    a = ap.parse_args()
    C, D, H, ps, k, B = 21, 384, 224, 16, 30, 64
    S = H // ps
    ext = TokenWorld(C, D, H, ps, seed=11)
    t0 = time.time()
    train = loader(a.train_images, B, H, C, seed=1000, with_255=True)
    val = loader(a.val_images, B, H, C, seed=9_000_000, with_255=True)
    ev = HbirdEvaluation(ext, train, num_classes=C, n_neighbours=k, device="cuda", nn_method="hip")
    t_build = time.time() - t0
    ext.log = []
    jac, det = ev.evaluate(val, S, return_knn_details=True, ignore_index=255)
    val_tok = np.concatenate(ext.log)
    ext.log = None
    fm, lm = ev.feature_memory.numpy(), ev.label_memory.numpy()
    q = val_tok.reshape(-1, D)
    # the engine's neighbours again, as indices (details carry features / labels only)
    idx_hip, dist_hip = ev.find_neighbours(torch.from_numpy(q).cuda(), k)
    idx_hip, dist_hip = idx_hip.cpu().numpy(), dist_hip.cpu().numpy()
    t1 = time.time()
    i32, d32 = oracle.knn_chain_f32(q, fm, k)
    t_chain = time.time() - t1
    t1 = time.time()
    i64, d64 = oracle.knn_f64(q, fm, k)
    t_f64 = time.time() - t1
    rep = oracle.near_tie_report(idx_hip, i64, d64)                       # 4 ulp of the score, SURVEY 8d
    rep64 = oracle.near_tie_report(idx_hip, i64, d64, ulps=64.0)           # a D = 384 fp32 chain is off by more than 4
    kf, kl = oracle.gather_neighbours(i64, fm, lm, val_tok.shape[0], S * S)
    lh_ref = oracle.cross_attention(val_tok, kf, kl)
    kf2, kl2 = oracle.gather_neighbours(i32, fm, lm, val_tok.shape[0], S * S)
    lh_same = oracle.cross_attention(val_tok, kf2, kl2)
    lh = det["knns_ca_labels"].numpy()
    m = oracle.PredsMIoUOracle(C, C, 255)
    for (x, y), n0 in zip(val, range(0, val_tok.shape[0], B)):
        gt = np.rint(y.numpy() * 255).astype(np.int64)
        m.update(gt, oracle.upsample_argmax(lh_ref[n0:n0 + gt.shape[0]], S, H, H))
    miou_ref = m.compute()[0]
    res = {
        "bank_rows": int(fm.shape[0]), "dim": D, "classes": C, "k": k, "queries": int(q.shape[0]),
        "vs_fp32_chain_oracle": {"indices_bit_exact": bool(np.array_equal(idx_hip, i32)),
                                 "distance_bits_equal": bool(np.array_equal(dist_hip.view(np.uint32), d32.view(np.uint32)))},
        "vs_float64_definition": rep, "vs_float64_definition_64ulp": rep64,
        "label_hat_max_abs_diff_oracle_on_same_neighbours": float(np.abs(lh - lh_same).max()),
        "label_hat_max_abs_diff_oracle_on_float64_neighbours": float(np.abs(lh - lh_ref).max()),
        "label_hat_rows_within_2e-5_of_float64_path": float((np.abs(lh - lh_ref).max(-1) < 2e-5).mean()),
        "miou_engine": float(jac), "miou_oracle": float(miou_ref), "miou_abs_delta": abs(float(jac) - float(miou_ref)),
        "seconds": {"bank_build_and_world": round(t_build, 1), "oracle_chain_f32": round(t_chain, 1), "oracle_f64": round(t_f64, 1)},
        "oracle_threads": oracle.num_threads(),
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
