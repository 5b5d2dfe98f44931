"""Dataset dispatch for `hbird_evaluation` (counterpart of hbird/data/__init__.py:39-198).

The reference's data pipeline (torchvision + pytorch_lightning datamodules, tar readers, random paired
transforms) is host-side I/O outside the accelerated hot path.  What the
evaluator needs from it is small: a train loader and a val loader yielding `(x [B,3,H,W] float,
y [B,1,H,W] float = mask/255)`, `get_train_dataset_size()`, `get_num_classes()` and the dataset's
ignore_index.  This module provides that contract for

  * "synthetic[*frac]"  -- a procedurally generated segmentation world (no files needed; used by the
                           smoke test, the end-to-end parity tests and the CLI's self-check), and
  * "voc", "ade20k", "cityscapes", "coco-thing", "coco-stuff" folder datasets via PIL + numpy
    (see folder.py) with the reference's class counts / ignore indices.
"""
from __future__ import annotations

from typing import Optional, Tuple

from .synthetic import SyntheticSegDataModule

# dataset -> (num_classes, ignore_index): voc_data.py:87-88, ade20k_data.py:51-52 + data/__init__.py:103,
# cityscapes_data.py:163-164, coco_data.py:73-77
DATASET_INFO = {
    "voc": (21, 255),
    "ade20k": (151, 0),
    "cityscapes": (19, 255),
    "coco-thing": (12, 255),
    "coco-stuff": (15, 255),
}


def get_dataset(dataset_name: str, data_dir: str, batch_size: int, num_workers: int, input_size,
                train_fs_path: Optional[str] = None, val_fs_path: Optional[str] = None) -> Tuple[object, int]:
    """Returns (datamodule, ignore_index).  `name*0.2` keeps that fraction of the training set
    (hbird/data/__init__.py:46-50).  `input_size` is the reference's square side, or an (H, W) frame size when the
    caller evaluates through sliding windows (hbird_mi/tiling.py)."""
    frac = 1.0
    name = dataset_name
    if "*" in dataset_name:
        name, f = dataset_name.split("*")
        frac = float(f)
    name = name.strip().lower()
    if name == "synthetic":
        dm = SyntheticSegDataModule(batch_size=batch_size, input_size=input_size, train_fraction=frac)
        return dm, 255
    if name in DATASET_INFO:
        from .folder import FolderSegDataModule
        num_classes, ignore = DATASET_INFO[name]
        dm = FolderSegDataModule(name, data_dir, batch_size, num_workers, input_size, num_classes, frac,
                                 train_fs_path, val_fs_path)
        return dm, ignore
    raise ValueError(f"Unknown dataset name: {dataset_name}")
