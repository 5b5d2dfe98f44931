"""CPU oracle (test infrastructure, NOT a product path).

ctypes front-end of oracle/hbird_oracle.c plus numpy restatements of the small integer/float
stages of the reference's hot path.  Every function cites the reference lines it follows
(paths relative to /root/reference).  The reference ships no tests or golden vectors for this
path, so the oracle is pinned against fixtures produced by importing the
reference's own Python in the build container: tests/golden/gen_golden.py -> tests/golden/*.npz,
checked by tests/test_oracle_golden.py.  The third-party Faiss-GPU arithmetic (exact flat
search) is restated from its definition; see hbird_oracle.c.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HBIRD_ORACLE_LIB") or os.path.join(_HERE, "libhbird_oracle.so")   # override: the sanitizer build (make asan)
_lib = None

__all__ = [
    "build", "lib", "knn_chain_f32", "knn_f64", "chain_sqnorm", "normalize_rows",
    "patchify_gt", "patch_label_hist", "cross_attention", "sample_patches", "sample_num_nonempty",
    "upsample_bilinear", "upsample_argmax", "confusion_matrix", "PredsMIoUOracle", "num_threads", "set_num_threads",
    "gather_neighbours", "near_tie_report", "window_origins", "sliding_window_argmax",
]

_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile oracle/hbird_oracle.c with gcc (called by __graft_entry__.build and lazily here)."""
    src = os.path.join(_HERE, "hbird_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if os.environ.get("HBIRD_ORACLE_LIB"):
        return _LIB_PATH          # an explicitly named build is used as it is
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libhbird_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_knn_chain_f32.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int64, _i64p, _f32p]
        L.orc_knn_f64.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int64, _i64p, _f64p]
        L.orc_chain_sqnorm.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.orc_normalize_rows.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.orc_patch_label_hist.argtypes = [_i64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, _f32p]
        L.orc_cross_attention.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_double, _f32p]
        for f in (L.orc_knn_chain_f32, L.orc_knn_f64, L.orc_chain_sqnorm, L.orc_normalize_rows,
                  L.orc_patch_label_hist, L.orc_cross_attention, L.orc_num_threads, L.orc_has_avx2):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    """OpenMP threads of the C restatement (bench.py's cpu_baseline: the cores the host grants, not the machine's count)."""
    L = lib()
    L.orc_set_num_threads.argtypes = [ctypes.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_set_num_threads(int(n))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


_METRIC = {"dot_product": 0, "ip": 0, "l2": 1, "euclidean": 1}


def knn_chain_f32(q, bank, k: int, metric: str = "dot_product", id_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Exact flat search in fp32 fmaf-chain arithmetic (the bit-exact target of the HIP kernel).

    Follows the contract of search_faiss.py:84-90: returns (indices int64 [nq,k], distances
    float32 [nq,k]), rows best-first (IP descending / squared-L2 ascending)."""
    q = _f32(q); bank = _f32(bank)
    nq, D = q.shape
    M = bank.shape[0]
    idx = np.empty((nq, k), dtype=np.int64)
    dist = np.empty((nq, k), dtype=np.float32)
    rc = lib().orc_knn_chain_f32(_p(q, _f32p), nq, _p(bank, _f32p), M, D, k, _METRIC[metric.lower()],
                                 id_base, _p(idx, _i64p), _p(dist, _f32p))
    if rc != 0:
        raise RuntimeError(f"orc_knn_chain_f32 failed rc={rc}")
    return idx, dist


def knn_f64(q, bank, k: int, metric: str = "dot_product", id_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Exact flat search, float64 definition of GpuIndexFlatIP / GpuIndexFlatL2 (search_faiss.py:43-46)."""
    q = _f32(q); bank = _f32(bank)
    nq, D = q.shape
    M = bank.shape[0]
    idx = np.empty((nq, k), dtype=np.int64)
    dist = np.empty((nq, k), dtype=np.float64)
    rc = lib().orc_knn_f64(_p(q, _f32p), nq, _p(bank, _f32p), M, D, k, _METRIC[metric.lower()], id_base,
                           _p(idx, _i64p), _p(dist, _f64p))
    if rc != 0:
        raise RuntimeError(f"orc_knn_f64 failed rc={rc}")
    return idx, dist


def chain_sqnorm(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty((x.shape[0],), dtype=np.float32)
    lib().orc_chain_sqnorm(_p(x, _f32p), x.shape[0], x.shape[1], _p(out, _f32p))
    return out


def normalize_rows(x) -> np.ndarray:
    """features / torch.norm(features, dim=-1, keepdim=True), no eps (hbird_eval.py:324, 335)."""
    x = _f32(x)
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    out = np.empty_like(x2)
    lib().orc_normalize_rows(_p(x2, _f32p), x2.shape[0], x2.shape[1], _p(out, _f32p))
    return out.reshape(shp)


def patchify_gt(gt: np.ndarray, patch_size: int) -> np.ndarray:
    """hbird_eval.py:555-573: [bs,c,h,w] -> [bs, h/ps, w/ps, c*ps*ps]."""
    bs, c, h, w = gt.shape
    g = gt.reshape(bs, c, h // patch_size, patch_size, w // patch_size, patch_size)
    g = g.transpose(0, 2, 4, 1, 3, 5)
    return g.reshape(bs, h // patch_size, w // patch_size, c * patch_size * patch_size)


def patch_label_hist(y: np.ndarray, patch_size: int, num_classes: int) -> np.ndarray:
    """Soft labels of hbird_eval.py:317-320 for y[B,1,H,W] (or [B,H,W]) int64 -> [B,S,S,C] fp32."""
    y = np.ascontiguousarray(y, dtype=np.int64)
    if y.ndim == 4:
        assert y.shape[1] == 1
        y = y[:, 0]
    B, H, W = y.shape
    S0, S1 = H // patch_size, W // patch_size
    out = np.empty((B, S0, S1, num_classes), dtype=np.float32)
    rc = lib().orc_patch_label_hist(_p(y, _i64p), B, H, W, patch_size, num_classes, _p(out, _f32p))
    if rc != 0:
        raise ValueError(f"orc_patch_label_hist rc={rc} (class id out of range or H/W not divisible)")
    return out


def cross_attention(q, k, v, beta: float = 0.02) -> np.ndarray:
    """hbird_eval.py:575-609 in float64: q[B,N,D], k[B,N,K,D], v[B,N,K,C] -> [B,N,C] fp32."""
    q = _f32(q); k = _f32(k); v = _f32(v)
    B, N, D = q.shape
    K = k.shape[2]
    C = v.shape[3]
    out = np.empty((B * N, C), dtype=np.float32)
    lib().orc_cross_attention(_p(q.reshape(B * N, D), _f32p), _p(k.reshape(B * N, K, D), _f32p),
                              _p(v.reshape(B * N, K, C), _f32p), B * N, K, D, C, float(beta), _p(out, _f32p))
    return out.reshape(B, N, C)


def gather_neighbours(idx: np.ndarray, feature_memory: np.ndarray, label_memory: np.ndarray, B: int, N: int):
    """hbird_eval.py:631-637: index_select of bank rows / label rows, viewed [B,N,k,-1]."""
    k = idx.shape[1]
    flat = idx.reshape(-1)
    return (feature_memory[flat].reshape(B, N, k, -1), label_memory[flat].reshape(B, N, k, -1))


def sample_num_nonempty(patchified_gts: np.ndarray, num_classes: int) -> np.ndarray:
    """Per-image count of non-empty patches = how many torch.rand values image b consumes
    (hbird_eval.py:490, 497-498)."""
    B = patchified_gts.shape[0]
    P = patchified_gts.shape[-1]
    g = patchified_gts.reshape(B, -1, P)
    ok = ((g >= 0) & (g < num_classes)).any(axis=2)
    return ok.sum(axis=1).astype(np.int64)


def sample_patches(patchified_gts: np.ndarray, num_classes: int, K: int, r: np.ndarray):
    """Bounded-memory patch sampling scores and choice (hbird_eval.py:447-517).

    patchified_gts [B,S,S,P] int64; r = the torch.rand(total_nz) draw (float32) consumed in image
    order (497-508).  Returns (sampled_indices [B,K] int64 ordered by ascending score like
    torch.topk(largest=False) -- ties broken by lower patch index --, scores [B,SS] fp32)."""
    B = patchified_gts.shape[0]
    P = patchified_gts.shape[-1]
    g = patchified_gts.reshape(B, -1, P).astype(np.int64)
    SS = g.shape[1]
    counts = np.zeros((B, SS, num_classes), dtype=np.int64)           # 476-478
    bi, pi = np.meshgrid(np.arange(B), np.arange(SS), indexing="ij")
    for p in range(P):
        np.add.at(counts, (bi, pi, g[:, :, p]), 1)
    presence = counts > 0                                              # 481
    class_freq = presence.sum(axis=1).astype(np.float32)               # 484
    scores = np.einsum("bpc,bc->bp", presence.astype(np.float32), class_freq).astype(np.float32)  # 489
    nonzero = presence.any(axis=2)                                     # 490
    scores[~nonzero] = np.float32(1e6)                                 # 493
    r = np.asarray(r, dtype=np.float32)
    rand_map = np.ones_like(scores)
    start = 0
    for b in range(B):                                                 # 497-507
        cnt = int(nonzero[b].sum())
        if cnt:
            rand_map[b, nonzero[b]] = r[start:start + cnt]
            start += cnt
    scores = (scores * rand_map).astype(np.float32)                    # 508
    order = np.argsort(scores, axis=1, kind="stable")[:, :K]           # 511 (ascending, lower index on ties)
    return order.astype(np.int64), scores


def _bilinear_axis(in_size: int, out_size: int):
    """ATen area_pixel_compute_source_index, align_corners=False, no scale_factor (fp32)."""
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0)).astype(np.float32)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1.0) - l1).astype(np.float32)
    return i0, i1, l0, l1


def upsample_bilinear(x: np.ndarray, h: int, w: int) -> np.ndarray:
    """F.interpolate(x[B,C,S0,S1].float(), size=(h,w), mode='bilinear') (hbird_eval.py:240)."""
    x = _f32(x)
    _, _, S0, S1 = x.shape
    y0, y1, ly0, ly1 = _bilinear_axis(S0, h)
    x0, x1, lx0, lx1 = _bilinear_axis(S1, w)
    top = x[:, :, y0][:, :, :, x0] * lx0 + x[:, :, y0][:, :, :, x1] * lx1
    bot = x[:, :, y1][:, :, :, x0] * lx0 + x[:, :, y1][:, :, :, x1] * lx1
    return (top * ly0[None, None, :, None] + bot * ly1[None, None, :, None]).astype(np.float32)


def upsample_argmax(label_hat: np.ndarray, S: int, h: int, w: int) -> np.ndarray:
    """hbird_eval.py:235-243: [B,N,C] -> reshape [B,S,S,C] -> permute -> bilinear -> argmax -> [B,1,h,w]."""
    B, N, C = label_hat.shape
    x = label_hat.reshape(B, S, S, C).transpose(0, 3, 1, 2)
    up = upsample_bilinear(x, h, w)
    return up.argmax(axis=1)[:, None].astype(np.int64)


def window_origins(H: int, W: int, win: int, stride: int):
    """Row-major window origins, last row / column flush with the border (sliding-window frames, cfg-5)."""
    def axis(n):
        o = list(range(0, n - win + 1, stride))
        if o[-1] != n - win:
            o.append(n - win)
        return o
    return [(y, x) for y in axis(H) for x in axis(W)]


def sliding_window_argmax(label_hats, origins, S: int, win: int, H: int, W: int):
    """Stitch the windows of a frame: every window's label_hat [B, S*S, C] is upsampled as hbird_eval.py:235-240
    does per image, summed (fp32, in the given order) into the frame, then argmax.  Returns (cluster_map
    [B,1,H,W] int64, acc [B,C,H,W] fp32)."""
    B, _, C = label_hats[0].shape
    acc = np.zeros((B, C, H, W), dtype=np.float32)
    for lh, (y0, x0) in zip(label_hats, origins):
        x = _f32(lh).reshape(B, S, S, C).transpose(0, 3, 1, 2)
        acc[:, :, y0:y0 + win, x0:x0 + win] += upsample_bilinear(x, win, win)
    return acc.argmax(axis=1)[:, None].astype(np.int64), acc


def confusion_matrix(gt: np.ndarray, pred: np.ndarray, num_gt: int, num_pred: int,
                     ignore_index: Optional[int]) -> np.ndarray:
    """PredsmIoU.update (hbird/utils/eval_metrics.py:73-104): rows = gt, cols = pred, int64."""
    gt = np.asarray(gt).reshape(-1).astype(np.int64)
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    if ignore_index is not None:
        m = gt != ignore_index
        gt, pred = gt[m], pred[m]
    valid = (gt >= 0) & (gt < num_gt) & (pred >= 0) & (pred < num_pred)
    gt, pred = gt[valid], pred[valid]
    return np.bincount(gt * num_pred + pred, minlength=num_gt * num_pred).reshape(num_gt, num_pred).astype(np.int64)


class PredsMIoUOracle:
    """numpy restatement of PredsmIoU.compute (hbird/utils/eval_metrics.py:112-288)."""

    def __init__(self, num_pred: int, num_gt: int, ignore_index: Optional[int] = None):
        self.num_pred, self.num_gt, self.ignore_index = int(num_pred), int(num_gt), ignore_index
        self.conf = np.zeros((self.num_gt, self.num_pred), dtype=np.int64)

    def update(self, gt, pred):
        self.conf += confusion_matrix(gt, pred, self.num_gt, self.num_pred, self.ignore_index)

    def score_matrix(self, precision_based: bool = False) -> np.ndarray:      # 112-131
        C = self.conf.astype(np.float64)
        row, col = C.sum(1, keepdims=True), C.sum(0, keepdims=True)
        if not precision_based:
            return C / np.maximum(row + col - C, 1e-8)
        return C / np.maximum(col, 1e-8)

    def compute(self, many_to_one: bool = False, precision_based: bool = False, linear_probe: bool = False):
        from scipy.optimize import linear_sum_assignment
        G, P = self.conf.shape
        if linear_probe:                                                      # 172-189
            tp, fp, fn = [], [], []
            col = self.conf.sum(0); row = self.conf.sum(1)
            for i in range(G):
                t = int(self.conf[i, i]) if i < P else 0
                tp.append(t); fp.append(int(col[i] - self.conf[i, i]) if i < P else 0); fn.append(int(row[i] - t))
            mapping = None
        else:
            if many_to_one:                                                   # 134-140
                mapping = self.score_matrix(precision_based).argmax(axis=0).astype(np.int64)
            else:                                                             # 143-159
                r, c = linear_sum_assignment(1.0 - self.score_matrix(False))
                mapping = np.zeros(P, dtype=np.int64)
                mapping[c] = r
            Cm = np.zeros((G, G), dtype=np.int64)                             # 197-198
            np.add.at(Cm, (slice(None), mapping), self.conf)
            tp_a = np.diag(Cm); fp_a = Cm.sum(0) - tp_a; fn_a = self.conf.sum(1) - tp_a
            tp, fp, fn = tp_a.tolist(), fp_a.tolist(), fn_a.tolist()
        tpd, fpd, fnd = (np.asarray(a, dtype=np.float64) for a in (tp, fp, fn))
        miou = float((tpd / np.maximum(tpd + fpd + fnd, 1e-8)).mean())        # 212-218
        return miou, tp, fp, fn, mapping


def near_tie_report(idx_test: np.ndarray, idx_ref64: np.ndarray, dist_ref64: np.ndarray, ulps: float = 4.0):
    """Parity of an fp32 search against the float64 definition, excusing near-ties.

    A position counts as excused when the float64 scores of the two rows involved differ by less
    than `ulps` fp32 ulps of the score magnitude.  Returns dict(ordered_rate, set_rate, excused_rate)."""
    nq, k = idx_ref64.shape
    ordered = (idx_test == idx_ref64)
    set_ok = np.array([set(idx_test[i]) == set(idx_ref64[i]) for i in range(nq)])
    tol = ulps * np.spacing(np.abs(dist_ref64).astype(np.float32)).astype(np.float64)
    gap_prev = np.abs(np.diff(dist_ref64, axis=1, prepend=np.inf))
    gap_next = np.abs(np.diff(dist_ref64, axis=1, append=-np.inf))
    near = (gap_prev < tol) | (gap_next < tol)
    excused = ordered | near
    return {
        "ordered_rate": float(ordered.all(axis=1).mean()),
        "set_rate": float(set_ok.mean()),
        "excused_rate": float(excused.all(axis=1).mean()),
        "pos_match": float(ordered.mean()),
    }
