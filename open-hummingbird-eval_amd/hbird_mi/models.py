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


def _nested_attr(obj, dotted: str) -> bool:
    for name in dotted.split("."):
        if not hasattr(obj, name):
            return False
        obj = getattr(obj, name)
    return True


class FeatureExtractor(nn.Module):
    """Auto-detecting extractor for DINO / DINOv2 / timm / HuggingFace ViTs (hbird/models.py:119-369): fp16 autocast +
    inference_mode, returns `(patch tokens [B,N,D] with the CLS token dropped, CLS attention | None)`.

    The backbone family is detected in the reference's order (models.py:326-354), because the families return
    DIFFERENT tokens for the same weights and the bank must be built from the ones the reference uses:
      dino    has get_intermediate_layers AND get_last_selfattention -> get_intermediate_layers(imgs)[0][:, 1:] (195-196)
      dinov2  class name contains "dino" and "v2", has forward_features -> dict entry 'x_norm_patchtokens' (199-206)
      timm    forward_features + blocks[0].attn -> forward_features(imgs)[:, 1:], i.e. tokens AFTER the final norm
              (208-216) -- NOT get_intermediate_layers, which timm >= 0.9 also has and which skips that norm by default
      hf      config.model_type in {vit, deit} -> last_hidden_state[:, 1:] (218-231)
      generic anything else: the module's output (token sequence, CLS dropped when present, or a [B,D,h,w] map)
    """

    def __init__(self, vit_model: nn.Module, eval_spatial_resolution: int = 14, d_model: int = 768,
                 use_autocast: bool = True, autocast_dtype: torch.dtype = torch.float16) -> None:
        super().__init__()
        self.model = vit_model
        self.eval_spatial_resolution = eval_spatial_resolution
        self.d_model = d_model
        self.use_autocast = use_autocast
        self.autocast_dtype = autocast_dtype
        self.backend = self._select_backend()

    def _select_backend(self) -> str:
        m = self.model
        if hasattr(m, "get_intermediate_layers") and hasattr(m, "get_last_selfattention"):
            return "dino"
        cls_name = type(m).__name__.lower()
        if hasattr(m, "forward_features") and "dino" in cls_name and "v2" in cls_name:
            return "dinov2"
        if hasattr(m, "forward_features") and _nested_attr(m, "blocks.0.attn"):
            return "timm"
        conf = getattr(m, "config", None)
        if conf is not None and str(getattr(conf, "model_type", "")).lower() in ("vit", "deit"):
            return "hf"
        return "generic"

    @staticmethod
    def _cls_attention(att: torch.Tensor) -> torch.Tensor:
        """[B, heads, N+1, N+1] -> CLS-to-patch attention averaged over heads, min-max normalised per image
        (models.py:44-54, 356-361)."""
        a = att[:, :, 0, 1:].mean(dim=1)
        lo, hi = a.min(dim=-1, keepdim=True).values, a.max(dim=-1, keepdim=True).values
        return (a - lo) / (hi - lo).clamp_min(1e-12)

    def _tokens(self, imgs):
        m = self.model
        if self.backend == "dino":
            return m.get_intermediate_layers(imgs)[0][:, 1:], self._cls_attention(m.get_last_selfattention(imgs))
        if self.backend == "dinov2":
            out = m.forward_features(imgs)
            return (out["x_norm_patchtokens"] if isinstance(out, dict) else out), None
        if self.backend == "timm":
            out = m.forward_features(imgs)
            if isinstance(out, dict):
                out = out.get("x", None) if out.get("x", None) is not None else (
                    out.get("tokens", None) if out.get("tokens", None) is not None else next(iter(out.values())))
            return out[:, 1:], None
        if self.backend == "hf":
            out = m(imgs, output_attentions=True, return_dict=True)
            att = self._cls_attention(out.attentions[-1]) if getattr(out, "attentions", None) else None
            return out.last_hidden_state[:, 1:], att
        # generic: whatever token stream the module offers
        if hasattr(m, "forward_features"):
            out = m.forward_features(imgs)
            if isinstance(out, dict):
                for key in ("x_norm_patchtokens", "patch_tokens", "last_hidden_state"):
                    if key in out:
                        out = out[key]
                        break
                else:
                    raise RuntimeError("FeatureExtractor: forward_features dict has no patch-token entry")
        elif hasattr(m, "get_intermediate_layers"):
            out = m.get_intermediate_layers(imgs)[0]
        else:
            out = m(imgs)
        if hasattr(out, "last_hidden_state"):
            out = out.last_hidden_state
        if out.dim() == 4:                                      # [B,D,h,w] feature map
            out = out.flatten(2).transpose(1, 2)
        n = out.shape[1]
        if int(math.isqrt(n)) ** 2 != n and int(math.isqrt(n - 1)) ** 2 == n - 1:
            out = out[:, 1:]                                    # drop CLS
        return out, None

    @torch.inference_mode()
    def forward_features(self, imgs: torch.Tensor):
        dev_type = imgs.device.type
        with torch.autocast(device_type=dev_type, dtype=self.autocast_dtype, enabled=(self.use_autocast and dev_type == "cuda")):
            tok, attn = self._tokens(imgs)
        return tok.float(), attn

    def forward(self, imgs: torch.Tensor):
        return self.forward_features(imgs)
