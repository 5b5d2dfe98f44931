#!/usr/bin/env python3
"""Command-line front end (counterpart of the reference's eval.py:369-505): same flags, plus `--nn-method hip`.

  python eval.py --dataset-name voc --data-dir /data/VOCSegmentation --d-model 384 --patch-size 16 \\
      --input-size 224 --batch-size 64 --device cuda --nn-method hip --checkpoint dino_vits16.pth --timm-model vit_small_patch16_224
  python eval.py --dataset-name synthetic --data-dir "" --d-model 3 --patch-size 8 --input-size 64 --device cuda   # self-check
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import random
import sys
import time
from typing import Any, Dict, List, Optional

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "open-hummingbird-eval_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _positive_int(v: str) -> int:
    i = int(v)
    if i <= 0:
        raise argparse.ArgumentTypeError("must be a positive integer")
    return i


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Hummingbird retrieval evaluation on MI355X (hbird_mi.hbird_evaluation).",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--dataset-name", required=True, help="voc | ade20k | cityscapes | synthetic, optionally 'name*0.2'")
    p.add_argument("--data-dir", required=True)
    p.add_argument("--d-model", type=_positive_int, required=True)
    p.add_argument("--patch-size", type=_positive_int, required=True)
    p.add_argument("--batch-size", type=_positive_int, default=64)
    p.add_argument("--input-size", type=_positive_int, default=224)
    p.add_argument("--frame-size", type=_positive_int, nargs=2, default=None, metavar=("H", "W"),
                   help="deliver H x W frames and process them through --input-size sliding windows (not in the reference)")
    p.add_argument("--window-stride", type=_positive_int, default=None, help="stride of the sliding windows (default: input size)")
    p.add_argument("--augmentation-epoch", type=_positive_int, default=1)
    p.add_argument("--num-workers", type=int, default=8)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--amp", action="store_true", help="accepted for compatibility (unused, as in the reference)")
    p.add_argument("--n-neighbours", type=_positive_int, default=30)
    p.add_argument("--nn-method", choices=["hip", "faiss", "scann"], default="hip")
    p.add_argument("--nn-param", action="append", default=[], metavar="KEY=VALUE")
    p.add_argument("--memory-size", type=int, default=None)
    p.add_argument("--ignore-index", type=int, default=255)
    p.add_argument("--train-fs", dest="train_fs_path", type=str, default=None)
    p.add_argument("--val-fs", dest="val_fs_path", type=str, default=None)
    p.add_argument("--timm-model", type=str, default=None)
    p.add_argument("--dinov2", type=str, choices=["vits14", "vitb14", "vitl14", "vitg14"], default=None)
    p.add_argument("--checkpoint", type=str, default=None)
    p.add_argument("--f-mem-p", type=str, default=None, help="feature-memory file: saved after the bank build; with --l-mem-p "
                   "and both files present the bank is loaded instead of rebuilt (the reference's f_mem_p, never reachable from its CLI)")
    p.add_argument("--l-mem-p", type=str, default=None, help="label-memory file (see --f-mem-p)")
    p.add_argument("--seed", type=int, default=123)
    p.add_argument("--out", type=str, default=None)
    p.add_argument("--log-level", choices=["DEBUG", "INFO", "WARNING", "ERROR"], default="INFO")
    return p


def parse_nn_params(kv_list: List[str]) -> Dict[str, Any]:
    """KEY=VALUE -> bool / int / float / str (reference eval.py:444-462)."""
    out: Dict[str, Any] = {}
    for kv in kv_list:
        if "=" not in kv:
            raise argparse.ArgumentTypeError(f"Invalid --nn-param '{kv}'. Use KEY=VALUE.")
        k, v = (s.strip() for s in kv.split("=", 1))
        if v.lower() in {"true", "false"}:
            out[k] = v.lower() == "true"
        else:
            for cast in (int, float):
                try:
                    out[k] = cast(v)
                    break
                except ValueError:
                    continue
            else:
                out[k] = v
        # a comma-separated list of GPU ids (`gpu_ids=0,1,2,3`; a single id has become an int above): the reference's parser has no
        # list form, its Faiss class iterates over whatever it gets (search_faiss.py:22)
        if k == "gpu_ids":
            val = out[k]
            out[k] = [val] if isinstance(val, int) else [int(t) for t in str(val).strip("[]").split(",") if t.strip()]
    return out


def set_seed(seed: Optional[int]) -> None:
    if seed is None:
        return
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class _PatchPool(torch.nn.Module):
    """Stand-in 'backbone' for the synthetic self-check: average-pools patches of the input image."""

    def __init__(self, patch):
        super().__init__()
        self.patch = patch

    def forward(self, x):
        return torch.nn.functional.avg_pool2d(x, self.patch).flatten(2).transpose(1, 2)


def build_model(args) -> torch.nn.Module:
    if args.dataset_name.split("*")[0] == "synthetic" and not (args.timm_model or args.dinov2):
        return _PatchPool(args.patch_size)
    if args.dinov2:
        model = torch.hub.load("facebookresearch/dinov2", f"dinov2_{args.dinov2}")          # eval.py:213 (needs a hub cache)
    elif args.timm_model:
        import timm
        model = timm.create_model(args.timm_model, pretrained=args.checkpoint is None, num_classes=0)
    else:
        raise SystemExit("one of --timm-model / --dinov2 is required for real datasets")
    if args.checkpoint:
        sd = torch.load(args.checkpoint, map_location="cpu")
        sd = sd.get("state_dict", sd.get("model", sd))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        logging.info("checkpoint loaded (missing %d, unexpected %d keys)", len(missing), len(unexpected))
    return model


def default_ftr_extr_fn(model, imgs):
    """Token extraction of the reference CLI (eval.py:262-309): prefer x_norm_patchtokens, else drop CLS."""
    out = model.forward_features(imgs) if hasattr(model, "forward_features") else model(imgs)
    if isinstance(out, dict):
        out = out.get("x_norm_patchtokens", next(iter(out.values())))
    if out.dim() == 3:
        n = out.shape[1]
        r = int(round((n - 1) ** 0.5))
        if int(round(n ** 0.5)) ** 2 != n and r * r == n - 1:
            out = out[:, 1:]
    return out, None


def main(argv: Optional[List[str]] = None) -> None:
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level), force=True)
    nn_params = parse_nn_params(args.nn_param)
    set_seed(args.seed)
    from hbird_mi.hbird_eval import hbird_evaluation
    model = build_model(args)
    t0 = time.time()
    result = hbird_evaluation(model, d_model=args.d_model, patch_size=args.patch_size, dataset_name=args.dataset_name,
                              data_dir=args.data_dir, batch_size=args.batch_size, input_size=args.input_size,
                              augmentation_epoch=args.augmentation_epoch, device=args.device, return_knn_details=False,
                              n_neighbours=args.n_neighbours, nn_method=args.nn_method, nn_params=nn_params,
                              ftr_extr_fn=default_ftr_extr_fn, memory_size=args.memory_size,
                              num_workers=args.num_workers, ignore_index=args.ignore_index,
                              train_fs_path=args.train_fs_path, val_fs_path=args.val_fs_path,
                              frame_size=tuple(args.frame_size) if args.frame_size else None,
                              window_stride=args.window_stride, f_mem_p=args.f_mem_p, l_mem_p=args.l_mem_p)
    from hbird_mi import hbird_eval as _he
    summary = {"miou": float(result), "seconds": round(time.time() - t0, 3), "nn_method": args.nn_method,
               "dataset": args.dataset_name, "n_neighbours": args.n_neighbours, **_he.last_run_info}
    print(json.dumps(summary))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f)


if __name__ == "__main__":
    main()
