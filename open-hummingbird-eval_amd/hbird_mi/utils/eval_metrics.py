"""`PredsmIoU` with the reference's API (hbird/utils/eval_metrics.py:13-339).

The streaming confusion matrix (the only per-pixel work) is the HIP kernel `hb_confusion_update`; the
O(C^2) tail -- IoU matrix, Hungarian / many-to-one mapping, TP/FP/FN, mIoU -- runs on the host in float64
exactly as the reference does it (scipy's linear_sum_assignment included, eval_metrics.py:154).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

from hbird_mi import ops

try:
    from scipy.optimize import linear_sum_assignment
    _SCIPY_AVAILABLE = True
except Exception:  # pragma: no cover
    _SCIPY_AVAILABLE = False


class PredsmIoU:
    def __init__(self, num_pred_classes: int, num_gt_classes: int, device: Optional[torch.device] = None,
                 ignore_index: Optional[int] = None, prefer_cuda: bool = True, store_reordered_preds: bool = True):
        self.num_pred_classes = int(num_pred_classes)
        self.num_gt_classes = int(num_gt_classes)
        self.ignore_index = int(ignore_index) if ignore_index is not None else None
        self.store_reordered_preds = bool(store_reordered_preds)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("PredsmIoU: no GPU visible; the confusion-matrix kernel has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PredsmIoU: device must be a GPU (the confusion-matrix kernel has no CPU fallback)")
        self._conf_mat = torch.zeros((self.num_gt_classes, self.num_pred_classes), dtype=torch.int64,
                                     device=self.device)
        self._pred_chunks: List[torch.Tensor] = []

    @torch.no_grad()
    def reset(self) -> None:
        self._conf_mat.zero_()
        self._pred_chunks.clear()

    @torch.no_grad()
    def update(self, gt: torch.Tensor, pred: torch.Tensor) -> None:
        if gt.shape != pred.shape:
            raise ValueError(f"Shapes must match. Got gt={gt.shape}, pred={pred.shape}")      # eval_metrics.py:78-79
        gt = gt.to(self.device, non_blocking=True).reshape(-1).long()
        pred = pred.to(self.device, non_blocking=True).reshape(-1).long()
        if gt.numel() == 0:
            return
        ops.confusion_update(self._conf_mat, gt, pred, self.ignore_index)
        if self.store_reordered_preds:
            # the reference keeps the predictions of the counted pixels (eval_metrics.py:106-109)
            keep = torch.ones_like(gt, dtype=torch.bool)
            if self.ignore_index is not None:
                keep &= gt.ne(self.ignore_index)
            keep &= (gt >= 0) & (gt < self.num_gt_classes) & (pred >= 0) & (pred < self.num_pred_classes)
            self._pred_chunks.append(pred[keep].to("cpu", dtype=torch.int32))

    @torch.no_grad()
    def update_from_label_hat(self, gt: torch.Tensor, label_hat: torch.Tensor, S: int) -> None:
        """`update(gt, upsample_argmax(label_hat))` without the class map (the fused K6 + K7 kernel): label_hat [B, S*S, C] soft
        predictions, gt [B,1,h,w].  The same confusion counts as the two-step path; with store_reordered_preds the map is kept."""
        if gt.numel() == 0:
            return
        gt = gt.to(self.device, non_blocking=True).long()
        pred = ops.upsample_argmax_confusion(label_hat, S, gt, self._conf_mat, self.ignore_index, want_map=self.store_reordered_preds)
        if self.store_reordered_preds:
            g, p = gt.reshape(-1), pred.reshape(-1)
            keep = torch.ones_like(g, dtype=torch.bool)
            if self.ignore_index is not None:
                keep &= g.ne(self.ignore_index)
            keep &= (g >= 0) & (g < self.num_gt_classes) & (p >= 0) & (p < self.num_pred_classes)
            self._pred_chunks.append(p[keep].to("cpu", dtype=torch.int32))

    # ---- O(C^2) host tail --------------------------------------------------------------------------
    def _conf_np(self) -> np.ndarray:
        return self._conf_mat.to("cpu").numpy().astype(np.int64)

    def _score_matrix(self, precision_based: bool = False) -> np.ndarray:                    # 112-131
        C = self._conf_np().astype(np.float64)
        row, col = C.sum(axis=1, keepdims=True), C.sum(axis=0, keepdims=True)
        if not precision_based:
            return C / np.maximum(row + col - C, 1e-8)
        return C / np.maximum(col, 1e-8)

    def _many_to_one_mapping(self, precision_based: bool = False) -> np.ndarray:             # 134-140
        return self._score_matrix(precision_based).argmax(axis=0).astype(np.int64)

    def _hungarian_mapping(self) -> np.ndarray:                                              # 143-159
        if not _SCIPY_AVAILABLE:
            raise RuntimeError("scipy is not available for Hungarian matching. Install scipy or use many_to_one=True.")
        row_ind, col_ind = linear_sum_assignment(1.0 - self._score_matrix(False))
        mapping = np.zeros(self.num_pred_classes, dtype=np.int64)       # unmatched predictions -> background 0
        mapping[col_ind] = row_ind
        return mapping

    def _tp_fp_fn_from_mapping(self, mapping: Optional[np.ndarray]):                         # 162-209
        C = self._conf_np()
        G, P = C.shape
        row_sum = C.sum(axis=1)
        if mapping is None:
            col_sum = C.sum(axis=0)
            tp, fp, fn = [], [], []
            for i in range(G):
                t = int(C[i, i]) if i < P else 0
                tp.append(t)
                fp.append(int(col_sum[i] - C[i, i]) if i < P else 0)
                fn.append(int(row_sum[i] - t))
            return tp, fp, fn
        Cm = np.zeros((G, G), dtype=np.int64)
        np.add.at(Cm, (slice(None), mapping), C)
        tp_t = np.diag(Cm)
        return tp_t.tolist(), (Cm.sum(axis=0) - tp_t).tolist(), (row_sum - tp_t).tolist()

    @staticmethod
    def _miou_from_counts(tp, fp, fn) -> float:                                              # 212-218
        tp_t, fp_t, fn_t = (np.asarray(a, dtype=np.float64) for a in (tp, fp, fn))
        return float((tp_t / np.maximum(tp_t + fp_t + fn_t, 1e-8)).mean())

    @torch.no_grad()
    def compute(self, is_global_zero: bool, many_to_one: bool = False, precision_based: bool = False,
                linear_probe: bool = False, sync_distributed: bool = False,
                return_reordered: bool = True) -> Tuple[float, List[int], List[int], List[int], List[int], float]:
        if not is_global_zero:
            return 0.0, [], [], [], [], 0.0                                                  # 246-248
        if sync_distributed and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(self._conf_mat, op=torch.distributed.ReduceOp.SUM)   # 251-252
        if linear_probe:
            mapping, bg = None, 0.0
        elif many_to_one:
            mapping = self._many_to_one_mapping(precision_based)
            bg = float((mapping == 0).sum() / max(self.num_pred_classes, 1))
        else:
            mapping = self._hungarian_mapping()
            bg = 1.0 / max(self.num_gt_classes, 1)
        tp, fp, fn = self._tp_fp_fn_from_mapping(mapping)
        miou = self._miou_from_counts(tp, fp, fn)
        if return_reordered:
            if not self.store_reordered_preds:
                raise RuntimeError("return_reordered=True requires store_reordered_preds=True during updates.")
            pred_all = torch.cat(self._pred_chunks, dim=0) if self._pred_chunks else torch.zeros(0, dtype=torch.int32)
            if mapping is None:
                reordered = pred_all.to(torch.int64)
            else:
                reordered = torch.from_numpy(mapping)[pred_all.to(torch.long)]
            reordered_list = reordered.to(torch.int64).tolist()
        else:
            reordered_list = []
        return miou, tp, fp, fn, reordered_list, bg
