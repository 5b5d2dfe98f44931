"""Device-side operators of the hot path: thin ctypes wrappers over the C ABI (include/hbird_hip.h).

All tensors are CUDA tensors on the current device; work is enqueued on torch's current stream.  torch is
used for memory and streams only -- every arithmetic step below runs in libhbird_hip.so.
"""
from __future__ import annotations

import ctypes

from typing import Optional

import torch

from hbird_mi import _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hbird_mi.ops: CUDA tensors required (the HIP path has no CPU fallback)")


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """x / ||x|| over the last dim, no eps (reference hbird_eval.py:324, 335)."""
    _need_cuda(x)
    x2 = x.contiguous().float().view(-1, x.shape[-1])
    out = torch.empty_like(x2)
    _lib.check(_lib.lib().hb_normalize_rows(_p(x2), x2.shape[0], x2.shape[1], _p(out), _stream(x2)))
    return out.view(x.shape)


def patch_label_hist(y: torch.Tensor, patch_size: int, num_classes: int, map255: bool = False) -> torch.Tensor:
    """y [B,1,H,W] int64 -> soft labels [B, H/ps, W/ps, C] (reference hbird_eval.py:310, 317-320)."""
    _need_cuda(y)
    if y.dim() != 4 or y.shape[1] != 1:
        raise ValueError(f"expected a [B,1,H,W] mask, got {tuple(y.shape)}")
    y = y.contiguous().to(torch.int64)
    B, _, H, W = y.shape
    out = torch.empty((B, H // patch_size, W // patch_size, num_classes), dtype=torch.float32, device=y.device)
    rc = _lib.lib().hb_patch_label_hist(_p(y), B, H, W, int(patch_size), int(num_classes), int(map255), _p(out), _stream(y))
    if rc != 0 and "out-of-range class" in _lib.last_error():
        raise _lib.HbirdClassRangeError(_lib.last_error())     # F.one_hot raises here (hbird_eval.py:319)
    _lib.check(rc)
    return out


def patch_scores(label: torch.Tensor):
    """label [B, SS, C] -> (scores [B,SS] fp32, nonempty [B,SS] int32, nz_count [B] int32)
    (reference hbird_eval.py:471-493)."""
    _need_cuda(label)
    label = label.contiguous()
    B, SS, C = label.shape
    scores = torch.empty((B, SS), dtype=torch.float32, device=label.device)
    nonempty = torch.empty((B, SS), dtype=torch.int32, device=label.device)
    nz = torch.empty((B,), dtype=torch.int32, device=label.device)
    _lib.check(_lib.lib().hb_patch_scores(_p(label), B, SS, C, _p(scores), _p(nonempty), _p(nz), _stream(label)))
    return scores, nonempty, nz


def patch_select(scores: torch.Tensor, nonempty: torch.Tensor, r: torch.Tensor, r_off: torch.Tensor, K: int,
                 want_scores: bool = False):
    """noise multiply + K smallest per image, ascending (reference hbird_eval.py:497-511)."""
    _need_cuda(scores, nonempty, r, r_off)
    B, SS = scores.shape
    out = torch.empty((B, K), dtype=torch.int64, device=scores.device)
    osc = torch.empty((B, SS), dtype=torch.float32, device=scores.device) if want_scores else None
    _lib.check(_lib.lib().hb_patch_select(_p(scores.contiguous()), _p(nonempty.contiguous()), _p(r.contiguous()),
                                          _p(r_off.contiguous()), B, SS, int(K), _p(out), _p(osc), _stream(scores)))
    return (out, osc) if want_scores else out


def gather_rows(src: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """src [R, W] fp32, ids [n] int64 -> [n, W]."""
    _need_cuda(src, ids)
    src = src.contiguous()
    ids = ids.contiguous().to(torch.int64).view(-1)
    out = torch.empty((ids.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().hb_gather_rows(_p(src), src.shape[0], src.shape[1], _p(ids), ids.numel(), _p(out),
                                         _stream(src)))
    return out


def upsample_argmax(label_hat: torch.Tensor, S: int, h: int, w: int) -> torch.Tensor:
    """label_hat [B, S*S, C] -> hard prediction [B,1,h,w] int64 (reference hbird_eval.py:235-243)."""
    _need_cuda(label_hat)
    label_hat = label_hat.contiguous().float()
    B, N, C = label_hat.shape
    if N != S * S:
        raise ValueError(f"label_hat has {N} patches, expected {S}x{S}")
    out = torch.empty((B, 1, h, w), dtype=torch.int64, device=label_hat.device)
    _lib.check(_lib.lib().hb_upsample_argmax(_p(label_hat), B, int(S), C, int(h), int(w), _p(out),
                                             _stream(label_hat)))
    return out


def upsample_argmax_confusion(label_hat: torch.Tensor, S: int, gt: torch.Tensor, conf: torch.Tensor, ignore_index,
                              want_map: bool = False) -> Optional[torch.Tensor]:
    """K6 + K7 fused: label_hat [B, S*S, C] upsampled to gt's [B,1,h,w], its argmax counted into conf [G,P] int64 against gt
    (reference hbird_eval.py:235-243 + eval_metrics.py:73-104).  The class map is only materialised when `want_map`."""
    _need_cuda(label_hat, gt, conf)
    label_hat = label_hat.contiguous().float()
    gt = gt.contiguous().to(torch.int64)
    B, N, C = label_hat.shape
    if N != S * S:
        raise ValueError(f"label_hat has {N} patches, expected {S}x{S}")
    if gt.dim() != 4 or gt.shape[0] != B or gt.shape[1] != 1:
        raise ValueError(f"gt must be [B,1,h,w] with B = {B}, got {tuple(gt.shape)}")
    h, w = int(gt.shape[2]), int(gt.shape[3])
    G, P = conf.shape
    has = ignore_index is not None
    out = torch.empty((B, 1, h, w), dtype=torch.int64, device=label_hat.device) if want_map else None
    _lib.check(_lib.lib().hb_upsample_argmax_confusion(_p(label_hat), B, int(S), C, h, w, _p(gt), G, P, int(ignore_index) if has else 0,
                                                       int(has), _p(conf), _p(out) if want_map else None, _stream(label_hat)))
    return out


def upsample_accumulate(label_hat: torch.Tensor, S: int, acc: torch.Tensor, y0: int, x0: int, win_h: int,
                        win_w: int) -> None:
    """Sliding windows: acc[B,H,W,C] (fp32, channels last) += bilinear upsample of one window's label_hat
    [B, S*S, C] to win_h x win_w, placed at (y0, x0) (the per-image upsample of hbird_eval.py:235-240)."""
    _need_cuda(label_hat)
    _need_cuda(acc)
    label_hat = label_hat.contiguous().float()
    B, N, C = label_hat.shape
    if N != S * S:
        raise ValueError(f"label_hat has {N} patches, expected {S}x{S}")
    if acc.dtype != torch.float32 or not acc.is_contiguous() or acc.dim() != 4 or acc.shape[0] != B or acc.shape[3] != C:
        raise ValueError("acc must be a contiguous float32 [B, H, W, C] tensor matching label_hat")
    _lib.check(_lib.lib().hb_upsample_accumulate(_p(label_hat), B, int(S), C, int(win_h), int(win_w), _p(acc),
                                                 acc.shape[1], acc.shape[2], int(y0), int(x0), _stream(label_hat)))


def argmax_channels(acc: torch.Tensor) -> torch.Tensor:
    """acc [B,H,W,C] float32 -> [B,1,H,W] int64 class map (first maximum wins, hbird_eval.py:243)."""
    _need_cuda(acc)
    acc = acc.contiguous().float()
    B, H, W, C = acc.shape
    out = torch.empty((B, 1, H, W), dtype=torch.int64, device=acc.device)
    _lib.check(_lib.lib().hb_argmax_channels(_p(acc), B * H * W, C, _p(out), _stream(acc)))
    return out


def confusion_update(conf: torch.Tensor, gt: torch.Tensor, pred: torch.Tensor, ignore_index) -> None:
    """conf [G,P] int64 (CUDA) += confusion counts (reference eval_metrics.py:73-104)."""
    _need_cuda(conf, gt, pred)
    gt = gt.contiguous().to(torch.int64).view(-1)
    pred = pred.contiguous().to(torch.int64).view(-1)
    if gt.numel() != pred.numel():
        raise ValueError("gt and pred must have the same number of elements")
    G, P = conf.shape
    has = ignore_index is not None
    _lib.check(_lib.lib().hb_confusion_update(_p(gt), _p(pred), gt.numel(), G, P, int(ignore_index) if has else 0,
                                              int(has), _p(conf), _stream(conf)))
