"""Deterministic synthetic inputs shared by tests/golden/gen_golden.py (which feeds them to the
reference's own Python) and by the tests (which feed the same inputs to the oracle / HIP path).

Pure numpy; nothing here reads /root/reference.
"""
from __future__ import annotations

import numpy as np


def unit_bank(M: int, D: int, seed: int) -> np.ndarray:
    """Rows of N(0,1), L2-normalised in float64 then rounded to fp32."""
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((M, D))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return b.astype(np.float32)


def vit_like_queries(nq: int, D: int, seed: int, scale: float = 3.0) -> np.ndarray:
    """Unnormalised queries (hbird_eval.py:625 sends un-normalised tokens to the backend)."""
    rng = np.random.default_rng(seed)
    return (scale * rng.standard_normal((nq, D))).astype(np.float32)


def random_masks(B: int, H: int, W: int, C: int, seed: int, with_255: bool = False) -> np.ndarray:
    """Rectangles of random classes on a background -> int64 [B,1,H,W]."""
    rng = np.random.default_rng(seed)
    y = np.zeros((B, 1, H, W), dtype=np.int64)
    for b in range(B):
        y[b, 0] = rng.integers(0, C)
        for _ in range(4):
            c = rng.integers(0, C)
            y0, x0 = rng.integers(0, H), rng.integers(0, W)
            y1, x1 = rng.integers(y0, H) + 1, rng.integers(x0, W) + 1
            y[b, 0, y0:y1, x0:x1] = c
        if with_255:
            y0, x0 = rng.integers(0, H - 2), rng.integers(0, W - 2)
            y[b, 0, y0:y0 + 2, x0:x0 + 3] = 255
    return y


def labels_from_masks(M: int, C: int, P: int, seed: int) -> np.ndarray:
    """Bank soft labels with values j/P (rows sum to 1), like hbird_eval.py:319-320 produces."""
    rng = np.random.default_rng(seed)
    cnt = rng.multinomial(P, np.ones(3) / 3.0, size=M)
    cls = rng.integers(0, C, size=(M, 3))
    lab = np.zeros((M, C), dtype=np.float64)
    for j in range(3):
        np.add.at(lab, (np.arange(M), cls[:, j]), cnt[:, j])
    return (lab / P).astype(np.float32)


class SegWorld:
    """Synthetic segmentation world: C class centroids in R^D; an image is a class map of
    rectangles; every pixel's D 'channels' are centroid[class] + noise.  A patch token is the mean
    over its pixels (the fake extractor), so tokens cluster by class and kNN label transfer works.
    """

    def __init__(self, C: int, D: int, H: int, ps: int, seed: int, noise: float = 0.5):
        self.C, self.D, self.H, self.ps, self.noise = C, D, H, ps, noise
        self.rng = np.random.default_rng(seed)
        self.centroids = self.rng.standard_normal((C, D)).astype(np.float32)

    def batch(self, B: int, with_255: bool = False):
        """Returns x [B,D,H,H] fp32 and y_float [B,1,H,H] = mask/255 (ToTensor convention)."""
        y = random_masks(B, self.H, self.H, self.C, int(self.rng.integers(1 << 30)), with_255=with_255)
        cls = np.where(y[:, 0] == 255, 0, y[:, 0])
        x = self.centroids[cls].transpose(0, 3, 1, 2)
        x = x + self.noise * self.rng.standard_normal(x.shape).astype(np.float32)
        return x.astype(np.float32), (y.astype(np.float32) / np.float32(255.0))

    def loader(self, n_batches: int, B: int, with_255: bool = False):
        return [self.batch(B, with_255) for _ in range(n_batches)]


def patch_mean_tokens(x: np.ndarray, ps: int) -> np.ndarray:
    """The fake extractor: x [B,D,H,W] -> tokens [B, (H/ps)*(W/ps), D] = per-patch pixel mean (fp32)."""
    B, D, H, W = x.shape
    t = x.reshape(B, D, H // ps, ps, W // ps, ps).astype(np.float32)
    t = t.mean(axis=(3, 5), dtype=np.float32)
    return np.ascontiguousarray(t.reshape(B, D, -1).transpose(0, 2, 1))
