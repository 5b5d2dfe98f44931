#!/usr/bin/env python3
"""End-to-end evaluator timing with a zero-cost token source: what the engine adds around the kernels.
usage: exp_e2e.py n_train n_val batch input patch D C [memory_size]   (E2E_FP16=1: nn_params use_fp16)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch
from hbird_mi.hbird_eval import HbirdEvaluation
from hbird_mi.models import FeatureExtractorSimple
from hbird_mi.data.synthetic import SyntheticSegDataModule
n_train, n_val, B, inp, ps, D, C = (int(x) for x in sys.argv[1:8])
mem = int(sys.argv[8]) if len(sys.argv) > 8 else None
S = inp // ps
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
def tokens(model, imgs):
    return torch.randn((imgs.shape[0], S * S, D), generator=g, device=dev)
t0 = time.time()
dm = SyntheticSegDataModule(batch_size=B, input_size=inp, num_classes=C, n_train=n_train, n_val=n_val)
print(f"synthetic data {time.time() - t0:.1f} s", flush=True)
ext = FeatureExtractorSimple(torch.nn.Identity(), tokens, eval_spatial_resolution=S, d_model=D)
train = dm.get_train_dataloader() if hasattr(dm, "get_train_dataloader") else dm.train_dataloader()
val = dm.get_val_dataloader() if hasattr(dm, "get_val_dataloader") else dm.val_dataloader()
torch.cuda.synchronize(); t0 = time.time()
kw = dict(memory_size=mem) if mem else {}
ev = HbirdEvaluation(ext, train, num_classes=C, n_neighbours=30, augmentation_epoch=1, device="cuda", nn_method="hip", dataset_size=n_train, nn_params={"use_fp16": bool(os.environ.get("E2E_FP16"))}, **kw)
torch.cuda.synchronize(); t1 = time.time()
print(f"bank build: {t1 - t0:.3f} s for {n_train} images ({(t1 - t0) / max(1, (n_train + B - 1) // B) * 1e3:.1f} ms per batch of {B})", flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t1 = time.time()
    jac = ev.evaluate(val, S, ignore_index=255)
    torch.cuda.synchronize(); t2 = time.time()
    nb = (n_val + B - 1) // B
    print(f"evaluate: {t2 - t1:.3f} s for {n_val} images = {(t2 - t1) / nb * 1e3:.1f} ms per batch of {B} ({B * S * S} query patches); mIoU {float(jac):.4f}", flush=True)
