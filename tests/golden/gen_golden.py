#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on fixed synthetic inputs.

Runs only in the build container (needs /root/reference, which is absent on the GPU box);
only the resulting data files are committed.  The reference imports torchvision and
pytorch_lightning at module scope (for its data pipeline, which is out of scope here); both are
absent from this image, so inert stub modules are registered before the import.  The third-party
faiss-gpu wheel is also absent: a stand-in `hbird.nn.search_faiss` module is registered whose
class derives from the reference's own NearestNeighborSearchBase and answers with the float64
definition of an exact flat search (the oracle's orc_knn_f64), so that the reference's
HbirdEvaluation(nn_method="faiss") + evaluate() run unmodified end to end.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvd = types.ModuleType("torchvision.datasets")

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x, *a, **k):
            return x

    class InterpolationMode:
        BILINEAR = "bilinear"
        NEAREST = "nearest"
        BICUBIC = "bicubic"

    for name in ("Compose", "ToTensor", "Normalize", "Resize", "ColorJitter", "RandomApply", "RandomGrayscale",
                 "GaussianBlur", "RandomResizedCrop", "RandomHorizontalFlip", "CenterCrop", "Lambda",
                 "RandomSolarize", "PILToTensor", "ConvertImageDtype"):
        setattr(tvt, name, type(name, (_Inert,), {}))
    tvt.InterpolationMode = InterpolationMode
    tvt.functional = tvf
    tvf.InterpolationMode = InterpolationMode
    tvd.VisionDataset = object
    tv.transforms = tvt
    tv.datasets = tvd
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.datasets": tvd})
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl


def _install_exact_backend():
    """hbird.nn.search_faiss stand-in: exact float64 flat search behind the reference's own ABC."""
    import oracle
    from hbird.nn.search_base import NearestNeighborSearchBase

    mod = types.ModuleType("hbird.nn.search_faiss")

    class NearestNeighborSearchFaiss(NearestNeighborSearchBase):
        def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", **kwargs):
            super().__init__(feature_memory, n_neighbors, distance_measure)

        def _initialize_index(self):
            return np.ascontiguousarray(self.feature_memory.cpu().numpy(), dtype=np.float32)

        def _add_features_to_index(self):
            pass

        def find_nearest_neighbors(self, q, k=None):
            k = self.n_neighbors if k is None else k
            idx, dist = oracle.knn_f64(q.cpu().numpy(), self.index, k, self.distance_measure)
            return idx, dist.astype(np.float32)

    mod.NearestNeighborSearchFaiss = NearestNeighborSearchFaiss
    sys.modules["hbird.nn.search_faiss"] = mod


class ReplayExtractor(torch.nn.Module):
    """Fake extractor: returns pre-computed tokens in call order (what hbird_eval.py:133,157,217,312 need)."""

    def __init__(self, tokens, eval_spatial_resolution, d_model):
        super().__init__()
        self.tokens = [torch.from_numpy(t) for t in tokens]
        self.eval_spatial_resolution = eval_spatial_resolution
        self.d_model = d_model
        self.i = 0

    def forward_features(self, x):
        t = self.tokens[self.i]
        self.i += 1
        return t.clone(), None


def main():
    _install_stubs()
    import golden_inputs as gi
    import hbird.hbird_eval as he
    from hbird.utils.eval_metrics import PredsmIoU
    import torch.nn.functional as F
    _install_exact_backend()
    out = os.environ.get("HBIRD_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # override: reproducibility check
    os.makedirs(out, exist_ok=True)
    HE = he.HbirdEvaluation
    blank = HE.__new__(HE)  # instance without running __init__, for the pure methods

    # ---- G1 patchify, G2 soft labels ------------------------------------------------------
    g = {}
    for name, (H, ps) in {"a": (28, 14), "b": (32, 16), "c": (56, 14)}.items():
        for C in (21, 151):
            y = gi.random_masks(2, H, H, C, seed=100 + H + C)
            pt = HE._patchify_gt(torch.from_numpy(y), ps)
            lab = F.one_hot(pt, num_classes=C).float().mean(dim=3)
            g[f"y_{name}_{C}"] = y
            g[f"ps_{name}_{C}"] = np.int64(ps)
            g[f"patches_{name}_{C}"] = pt.numpy()
            g[f"label_{name}_{C}"] = lab.numpy()
    np.savez_compressed(os.path.join(out, "g12_patchify_softlabels.npz"), **g)

    # ---- G3 cross attention -----------------------------------------------------------------
    g = {}
    rng = np.random.default_rng(3)
    for name, (B, N, K, D, C) in {"small": (2, 9, 5, 16, 4), "vitS": (2, 9, 30, 384, 21)}.items():
        q = (3 * rng.standard_normal((B, N, D))).astype(np.float32)
        k = rng.standard_normal((B, N, K, D)).astype(np.float32)
        k /= np.linalg.norm(k, axis=-1, keepdims=True)
        # make the neighbours correlated with q so the softmax is not one-hot noise
        k = (k + 0.15 * q[:, :, None, :] / np.linalg.norm(q, axis=-1)[:, :, None, None]).astype(np.float32)
        v = gi.labels_from_masks(B * N * K, C, 196, seed=31).reshape(B, N, K, C)
        o = blank._cross_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v))
        g.update({f"q_{name}": q, f"k_{name}": k, f"v_{name}": v, f"out_{name}": o.numpy()})
    np.savez_compressed(os.path.join(out, "g3_cross_attention.npz"), **g)

    # ---- G4 exact kNN through _find_nearest_key_to_query --------------------------------------
    g = {}
    for name, (M, D, B, N, k, C, metric) in {
        "ip32": (4096, 32, 2, 16, 30, 21, "dot_product"),
        "l2_32": (4096, 32, 2, 16, 30, 21, "l2"),
        "ip384": (20000, 384, 2, 49, 30, 21, "dot_product"),
    }.items():
        bank = gi.unit_bank(M, D, seed=41)
        lab = gi.labels_from_masks(M, C, 196, seed=42)
        q = gi.vit_like_queries(B * N, D, seed=43).reshape(B, N, D)
        if name == "ip32":
            bank[1234] = bank[77]      # constructed exact ties: duplicate rows ...
            bank[4000] = bank[77]
            q[0, 0] = 5.0 * bank[77]   # ... that are certainly among the neighbours of query 0
        ev = HE.__new__(HE)
        ev.feature_memory = torch.from_numpy(bank)
        ev.label_memory = torch.from_numpy(lab)
        ev.n_neighbours = k
        ev._create_nn(k, nn_method="faiss", distance_measure=metric)
        idx, dist = ev.NN_algorithm.find_nearest_neighbors(torch.from_numpy(q.reshape(B * N, D)))
        kf, kl = ev._find_nearest_key_to_query(torch.from_numpy(q))
        g.update({f"shape_{name}": np.array([M, D, B, N, k, C]), f"metric_{name}": np.array(metric),
                  f"idx_{name}": idx, f"dist_{name}": dist,
                  f"kl_{name}": kl.numpy(), f"kf_rowsum_{name}": kf.numpy().sum(-1)})
    np.savez_compressed(os.path.join(out, "g4_knn.npz"), **g)

    # ---- G5 _sample_features -------------------------------------------------------------------
    g = {}
    for name, (B, H, ps, D, C, K, seed) in {"a": (3, 32, 8, 8, 6, 5, 5), "b": (2, 56, 14, 16, 21, 7, 6)}.items():
        y = gi.random_masks(B, H, H, C, seed=50 + seed)
        S = H // ps
        feats = np.random.default_rng(seed).standard_normal((B, S * S, D)).astype(np.float32)
        pt = HE._patchify_gt(torch.from_numpy(y), ps)
        ev = HE.__new__(HE)
        ev.num_sampled_features = K
        torch.manual_seed(seed)
        state = torch.get_rng_state()
        sf, si = ev._sample_features(torch.from_numpy(feats), pt, C)
        torch.set_rng_state(state)
        r = torch.rand(S * S * B)  # every patch is non-empty for in-range masks -> total_nz = B*S*S
        g.update({f"y_{name}": y, f"feats_{name}": feats, f"cfg_{name}": np.array([ps, C, K, seed]),
                  f"r_{name}": r.numpy(), f"sidx_{name}": si.numpy(), f"sfeat_{name}": sf.numpy()})
    np.savez_compressed(os.path.join(out, "g5_sample.npz"), **g)

    # ---- G6 _create_memory (unbounded / bounded / bounded+trim) and G7 full evaluate -----------
    g = {}
    for name, (C, D, H, ps, nb, B, k, mem, aug, ign) in {
        "unb": (5, 16, 32, 8, 4, 3, 5, None, 1, 255),
        "bnd": (5, 16, 32, 8, 4, 3, 5, 60, 1, 255),      # 60 // 12 = 5 per image -> 60 rows, no trim
        "trim": (21, 32, 56, 14, 3, 4, 8, 100, 2, 255),  # 100 // 24 = 4 per image -> 96 rows -> trimmed
        "ade": (21, 32, 56, 14, 3, 4, 8, None, 1, 0),    # ignore_index 0 variant
    }.items():
        world = gi.SegWorld(C, D, H, ps, seed=70 + len(name))
        train = world.loader(nb, B, with_255=True)
        val = world.loader(2, B, with_255=(ign == 255))
        S = H // ps
        tr_tok = [gi.patch_mean_tokens(x, ps) for x, _ in train] * aug
        va_tok = [gi.patch_mean_tokens(x, ps) for x, _ in val]
        ext = ReplayExtractor(tr_tok + va_tok, S, D)
        tl = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in train]
        vl = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in val]
        torch.manual_seed(1234)
        state = torch.get_rng_state()
        ev = HE(ext, tl, num_classes=C, n_neighbours=k, augmentation_epoch=aug, device="cpu",
                nn_method="faiss", nn_params={}, memory_size=mem, dataset_size=nb * B)
        jac, det = ev.evaluate(vl, eval_spatial_resolution=S, return_knn_details=True, ignore_index=ign)
        # replay of the evaluate loop to also record the hard predictions (hbird_eval.py:235-243)
        lh = det["knns_ca_labels"]
        bs = lh.shape[0]
        up = F.interpolate(lh.reshape(bs, S, S, C).permute(0, 3, 1, 2).float(), size=(H, H), mode="bilinear")
        cmap = up.argmax(dim=1).unsqueeze(1)
        g.update({
            f"cfg_{name}": np.array([C, D, H, ps, nb, B, k, -1 if mem is None else mem, aug, ign]),
            f"rng_state_{name}": state.numpy(),
            f"feature_memory_{name}": ev.feature_memory.numpy(), f"label_memory_{name}": ev.label_memory.numpy(),
            f"jac_{name}": np.float64(jac), f"knns_labels_{name}": det["knns_labels"].numpy(),
            f"knns_rowsum_{name}": det["knns"].numpy().sum(-1),
            f"label_hat_{name}": lh.numpy(), f"cluster_map_{name}": cmap.numpy().astype(np.uint8),
        })
        if name == "unb":
            g[f"upsampled_{name}"] = up.numpy()
        for i, (x, y) in enumerate(train):
            g[f"train_y_{name}_{i}"] = y
            g[f"train_tok_{name}_{i}"] = tr_tok[i]
        for i, (x, y) in enumerate(val):
            g[f"val_y_{name}_{i}"] = y
            g[f"val_tok_{name}_{i}"] = va_tok[i]
    np.savez_compressed(os.path.join(out, "g67_memory_evaluate.npz"), **g)

    # ---- G8 PredsmIoU ---------------------------------------------------------------------------
    g = {}
    rng = np.random.default_rng(8)
    for name, (C, n, ign) in {"c5": (5, 4000, 255), "c21": (21, 30000, 255), "ade": (21, 30000, 0)}.items():
        gt = rng.integers(0, C, size=n)
        pred = np.where(rng.random(n) < 0.7, (gt * 3 + 1) % C, rng.integers(0, C, size=n))  # permuted + noise
        gt[rng.random(n) < 0.05] = ign
        gt[:3] = C + 7          # out-of-range values are dropped (eval_metrics.py:92-95)
        pred[3:5] = -1
        for mode, kw in {"hung": {}, "m2o": {"many_to_one": True}, "m2o_prec": {"many_to_one": True, "precision_based": True},
                         "lin": {"linear_probe": True}}.items():
            m = PredsmIoU(C, C, ignore_index=ign, device=torch.device("cpu"))
            m.update(torch.from_numpy(gt.reshape(2, -1)), torch.from_numpy(pred.reshape(2, -1)))
            miou, tp, fp, fn, reordered, bg = m.compute(is_global_zero=True, **kw)
            g.update({f"miou_{name}_{mode}": np.float64(miou), f"tp_{name}_{mode}": np.array(tp),
                      f"fp_{name}_{mode}": np.array(fp), f"fn_{name}_{mode}": np.array(fn),
                      f"bg_{name}_{mode}": np.float64(bg), f"nreordered_{name}_{mode}": np.int64(len(reordered)),
                      f"reordered_head_{name}_{mode}": np.array(reordered[:64])})
        g.update({f"gt_{name}": gt, f"pred_{name}": pred, f"cfg_{name}": np.array([C, n, ign]),
                  f"conf_{name}": m._conf_mat.numpy()})
    np.savez_compressed(os.path.join(out, "g8_predsmiou.npz"), **g)

    # ---- G9 the reference FeatureExtractor's generic QKV-hook path on a tiny random ViT (models.py:257-321) ------------
    # forward_features() of the generic backend asks for layer -1 through a dotted-name lookup ("blocks.-1.attn.qkv") that an
    # nn.ModuleList never resolves: it raises RuntimeError for every model (recorded below).  With an explicit non-negative layer
    # the hook path itself works, for a qkv module with 5-D output: that pins the unpacking (reshape / permute / CLS drop).
    from hbird.models import FeatureExtractor as RefFE
    from tiny_vit import TinyQKVViT
    g = {}
    vit = TinyQKVViT(seed=5).eval()
    imgs = torch.from_numpy(np.random.default_rng(77).standard_normal((2, 3, 32, 32)).astype(np.float32))
    fe = RefFE(vit, eval_spatial_resolution=4, d_model=16)
    g["backend"] = np.array(fe._backend.name)
    g["imgs"] = imgs.numpy()
    try:
        fe.forward_features(imgs)
        g["default_layer_raises"] = np.array("")
    except RuntimeError as e:
        g["default_layer_raises"] = np.array(type(e).__name__)
    for feat in ("q", "k", "v"):
        for layer in (0, 1):
            f, att = fe.get_intermediate_layer_feats(imgs, feat=feat, layer_num=layer)
            assert att is None
            g[f"feats_{feat}_{layer}"] = f.numpy()
    fe.freeze_feature_extractor(["blocks.1.attn"])
    g["trainable_after_freeze"] = np.array(sorted(n for n, p_ in vit.named_parameters() if p_.requires_grad))
    fe.freeze_feature_extractor([r"blocks\.0\..*bias$"], regex=True)
    g["trainable_after_regex_freeze"] = np.array(sorted(n for n, p_ in vit.named_parameters() if p_.requires_grad))
    np.savez_compressed(os.path.join(out, "g9_feature_extractor.npz"), **g)

    # ---- G10 _create_memory over training batches of TWO input sizes (hbird_eval.py:313-314 recomputes patch_size per batch) ------------
    # Same eval_spatial_resolution S = 4, inputs of 32 px (patch 8 x 8: labels j / 64) and 64 px (patch 16 x 16: labels j / 256), unbounded
    # bank; then an evaluation on 32-px validation batches.
    g = {}
    C, D, S, B, k = 5, 16, 4, 3, 5
    wa, wb = gi.SegWorld(C, D, 32, 8, seed=91), gi.SegWorld(C, D, 64, 16, seed=92)
    wb.centroids = wa.centroids.copy()                       # one world, two resolutions
    train = [wa.batch(B, True), wb.batch(B, True), wa.batch(B, True), wb.batch(B, True)]
    val = wa.loader(2, B, with_255=True)
    tr_tok = [gi.patch_mean_tokens(x, x.shape[-1] // S) for x, _ in train]
    va_tok = [gi.patch_mean_tokens(x, 8) for x, _ in val]
    ext = ReplayExtractor(tr_tok + va_tok, S, D)
    tl = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in train]
    vl = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in val]
    torch.manual_seed(4321)
    ev = HE(ext, tl, num_classes=C, n_neighbours=k, augmentation_epoch=1, device="cpu", nn_method="faiss", nn_params={}, memory_size=None)
    jac = ev.evaluate(vl, eval_spatial_resolution=S, ignore_index=255)
    g.update({"cfg": np.array([C, D, S, B, k]), "feature_memory": ev.feature_memory.numpy(), "label_memory": ev.label_memory.numpy(),
              "jac": np.float64(jac)})
    for i, (x, y) in enumerate(train):
        g[f"train_y_{i}"] = y; g[f"train_tok_{i}"] = tr_tok[i]
    for i, (x, y) in enumerate(val):
        g[f"val_y_{i}"] = y; g[f"val_tok_{i}"] = va_tok[i]
    np.savez_compressed(os.path.join(out, "g10_mixed_patch_sizes.npz"), **g)

    # (y/255)*255 round trip that hbird_eval.py:219,309 rely on
    c = np.arange(256, dtype=np.float32)
    rt = (torch.from_numpy(c / np.float32(255.0)) * 255).long().numpy()
    assert (rt == np.arange(256)).all(), "mask/255*255 does not round-trip"
    tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print("golden fixtures written to", out, "total bytes", tot)


if __name__ == "__main__":
    main()
