"""Thin counterparts of the reference's extractor wrappers (hbird/models.py:70-103, 119-369).

The ViT forward itself is ordinary PyTorch-ROCm and is not part of the accelerated path; the evaluator only
needs `forward_features(imgs) -> (tokens [B,N,D], attn|None)`, `eval_spatial_resolution` and `d_model`
(read at hbird_eval.py:133, 157).
"""
from __future__ import annotations

import math
from typing import Callable

import torch
import torch.nn as nn


class FeatureExtractorSimple(nn.Module):
    """User-supplied token extraction: `ftr_extr_fn(model, imgs)` returns `tokens [B,N,D]` or `(tokens, attn)`
    (counterpart of hbird/models.py:70-103; this is what the CLI always uses, eval.py:324)."""

    def __init__(self, vit_model: nn.Module, ftr_extr_fn: Callable, eval_spatial_resolution: int = 14,
                 d_model: int = 768) -> None:
        super().__init__()
        self.model, self.ftr_extr_fn = vit_model, ftr_extr_fn
        self.eval_spatial_resolution, self.d_model = eval_spatial_resolution, d_model

    def forward_features(self, imgs: torch.Tensor):
        out = self.ftr_extr_fn(self.model, imgs)
        return out if isinstance(out, (tuple, list)) else (out, None)

    forward = forward_features


class FeatureExtractor(nn.Module):
    """Auto-detecting extractor for DINO / DINOv2 / timm-style ViTs (hbird/models.py:164-235): fp16 autocast +
    inference_mode, returns patch tokens with the CLS token dropped and `None` for the attention map."""

    def __init__(self, vit_model: nn.Module, eval_spatial_resolution: int = 14, d_model: int = 768) -> None:
        super().__init__()
        self.model = vit_model
        self.eval_spatial_resolution = eval_spatial_resolution
        self.d_model = d_model

    def _tokens(self, imgs):
        m = self.model
        if hasattr(m, "get_intermediate_layers"):               # DINO (models.py:195-196) / DINOv2
            out = m.get_intermediate_layers(imgs)[0]
            n = self.eval_spatial_resolution ** 2
            return out[:, -n:] if out.shape[1] > n else out
        if hasattr(m, "forward_features"):
            out = m.forward_features(imgs)
            if isinstance(out, dict):                           # DINOv2 (models.py:201-203)
                for key in ("x_norm_patchtokens", "patch_tokens", "last_hidden_state"):
                    if key in out:
                        out = out[key]
                        break
                else:
                    raise RuntimeError("FeatureExtractor: forward_features dict has no patch-token entry")
        else:
            out = m(imgs)
        if hasattr(out, "last_hidden_state"):
            out = out.last_hidden_state
        if out.dim() == 4:                                      # [B,D,h,w] feature map
            out = out.flatten(2).transpose(1, 2)
        n = out.shape[1]
        if int(math.isqrt(n)) ** 2 != n and int(math.isqrt(n - 1)) ** 2 == n - 1:
            out = out[:, 1:]                                    # drop CLS
        return out

    @torch.inference_mode()
    def forward_features(self, imgs: torch.Tensor):
        dev_type = imgs.device.type
        with torch.autocast(device_type=dev_type, dtype=torch.float16, enabled=(dev_type == "cuda")):
            tok = self._tokens(imgs)
        return tok.float(), None

    def forward(self, imgs: torch.Tensor):
        return self.forward_features(imgs)
