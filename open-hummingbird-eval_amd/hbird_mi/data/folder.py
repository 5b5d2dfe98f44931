"""Folder datasets for VOC / ADE20K / Cityscapes / COCO (thing, stuff) with paired (image, mask) transforms, PIL + numpy + torch only
(counterparts of hbird/data/{voc,ade20k,cityscapes}/*_data.py and hbird/utils/{transforms,
image_transformations}.py; host-side I/O, "next" row f3 of SURVEY.md section 8).

The dataset root may be a directory or a .tar archive (`/data/voc.tar`, `/data/all.tar!/VOC`, as the reference's
hbird/utils/io.py and voc_tar_data.py accept).  Directory conventions follow the reference (DATASET.md):
  voc         <root>/images/*.jpg, <root>/SegmentationClassAug/*.png (train), <root>/SegmentationClass/*.png (val),
              <root>/sets/{trainaug,val}.txt                                   (voc_data.py:137-160)
  ade20k      <root>/images/{training,validation}/*.jpg, <root>/annotations/{training,validation}/*.png
                                                                               (ade20k_data.py:72-88)
  cityscapes  <root>/leftImg8bit/{train,val}/<city>/*_leftImg8bit.png,
              <root>/gtFine/{train,val}/<city>/*_gtFine_labelIds.png, labelId -> trainId (cityscapes_data.py:28-48)
  coco-*      <root>/images/{train,val}2017/*.jpg, <root>/annotations/... pixel maps + category json (coco_data.py:95-146)
Samples are `(x float [3,S,S] normalised, y float [1,S,S] = mask / 255)` exactly as the reference's
CombTransforms + ToTensor deliver them (hence `y * 255` in the evaluator, hbird_eval.py:219, 309).
"""
from __future__ import annotations

import math
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils.data import DataLoader, Dataset

IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.255]      # sic: the reference's value (hbird/utils/transforms.py:29)


class DiskStore:
    """A dataset root on disk; paths are relative to it."""

    def __init__(self, root: str):
        self.root = root

    def isdir(self, rel: str) -> bool:
        return os.path.isdir(os.path.join(self.root, rel))

    def listdir(self, rel: str) -> List[str]:
        return sorted(os.listdir(os.path.join(self.root, rel)))

    def open(self, rel: str):
        return open(os.path.join(self.root, rel), "rb")


class TarStore:
    """A dataset root inside a .tar archive: `/data/voc.tar` or `/data/all.tar!/VOC/` (the notation of the
    reference's hbird/utils/io.py:10-15 and voc_tar_data.py).  The member table is read once; the archive itself
    is opened lazily per process, because a TarFile handle must not be shared with DataLoader workers
    (voc_tar_data.py:228-231)."""

    def __init__(self, path: str):
        tar_path, _, inner = path.partition(".tar")
        for ext in (".gz", ".bz2", ".xz"):
            if inner.startswith(ext):
                tar_path, inner = tar_path + ".tar" + ext, inner[len(ext):]
                break
        else:
            tar_path += ".tar"
        if not os.path.isfile(tar_path):
            raise FileNotFoundError(f"Tar archive not found: {tar_path}")                # io.py:34-35
        self.tar_path = tar_path
        self.prefix = inner.lstrip("!").strip("/")
        self._tar, self._pid = None, None
        self.files, self.dirs = {}, {""}
        import tarfile
        with tarfile.open(tar_path, "r:*") as t:
            for m in t.getmembers():
                name = os.path.normpath(m.name).lstrip("./")
                if self.prefix:
                    if not name.startswith(self.prefix + "/"):
                        continue
                    name = name[len(self.prefix) + 1:]
                if m.isfile():
                    self.files[name] = m
                    d = os.path.dirname(name)
                    while d and d not in self.dirs:
                        self.dirs.add(d)
                        d = os.path.dirname(d)
                elif m.isdir():
                    self.dirs.add(name.rstrip("/"))

    def isdir(self, rel: str) -> bool:
        return rel.strip("/") in self.dirs

    def listdir(self, rel: str) -> List[str]:
        rel = rel.strip("/")
        pre = rel + "/" if rel else ""
        names = {n[len(pre):].split("/", 1)[0] for n in list(self.files) + list(self.dirs) if n.startswith(pre) and n != rel}
        return sorted(n for n in names if n)

    def open(self, rel: str):
        import io
        import tarfile
        if self._tar is None or self._pid != os.getpid():
            self._tar, self._pid = tarfile.open(self.tar_path, "r:*"), os.getpid()
        if rel not in self.files:
            raise FileNotFoundError(f"File '{rel}' not found inside {self.tar_path}")     # io.py:91
        return io.BytesIO(self._tar.extractfile(self.files[rel]).read())

    def __getstate__(self):           # DataLoader workers (spawn) get a copy without the open handle
        d = dict(self.__dict__)
        d["_tar"], d["_pid"] = None, None
        return d


def open_store(root: str):
    return TarStore(root) if ".tar" in root else DiskStore(root)


def read_file_set(path: str) -> List[str]:
    """Lines of a split file, on disk or inside an archive (`/x/a.tar!/sets/val.txt`, io.py:60-103)."""
    if ".tar" in path:
        head, _, inner = path.partition("!")
        if inner:
            with TarStore(head).open(inner.strip("/")) as f:
                return [ln.strip() for ln in f.read().decode().splitlines() if ln.strip()]
    with open(path) as f:
        return [ln.strip() for ln in f if ln.strip()]


def _to_tensor_img(img: Image.Image) -> torch.Tensor:
    a = np.asarray(img, dtype=np.float32) / 255.0
    t = torch.from_numpy(a).permute(2, 0, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (t - mean) / std


def _to_tensor_mask(mask: Image.Image) -> torch.Tensor:
    a = np.asarray(mask, dtype=np.uint8)
    return torch.from_numpy(a.astype(np.float32) / np.float32(255.0)).unsqueeze(0)


def _wh(size) -> Tuple[int, int]:
    """PIL (width, height) of an int (square, as in the reference) or an (H, W) frame size (sliding windows)."""
    return (size, size) if isinstance(size, int) else (int(size[1]), int(size[0]))


class ValTransform:
    """Resize both to (S,S): bilinear for the image, nearest for the mask (transforms.py:215-236)."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img, mask):
        img = img.resize(_wh(self.size), Image.BILINEAR)
        mask = mask.resize(_wh(self.size), Image.NEAREST)
        return _to_tensor_img(img), _to_tensor_mask(mask)


class TrainTransform:
    """Colour jitter on the image (each of brightness/contrast/saturation/hue with p = 0.5, range 0.1) then a
    shared RandomResizedCrop(size, scale=(0.5, 2.0), ratio=(3/4, 4/3)) (transforms.py:166-212)."""

    def __init__(self, size, jitter: float = 0.1, p: float = 0.5, scale=(0.5, 2.0), ratio=(3 / 4, 4 / 3)):
        self.size, self.jitter, self.p, self.scale, self.ratio = size, jitter, p, scale, ratio

    def _jitter(self, img):
        j = self.jitter
        if random.random() < self.p:
            img = ImageEnhance.Brightness(img).enhance(random.uniform(1 - j, 1 + j))
        if random.random() < self.p:
            img = ImageEnhance.Contrast(img).enhance(random.uniform(1 - j, 1 + j))
        if random.random() < self.p:
            img = ImageEnhance.Color(img).enhance(random.uniform(1 - j, 1 + j))
        if random.random() < self.p:
            h, s, v = img.convert("HSV").split()
            shift = int(round(random.uniform(-j, j) * 255))
            h = h.point(lambda x: (x + shift) % 256)
            img = Image.merge("HSV", (h, s, v)).convert("RGB")
        return img

    def _crop_box(self, w, h):
        area = w * h
        log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * random.uniform(*self.scale)
            ar = math.exp(random.uniform(*log_r))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                x0, y0 = random.randint(0, w - cw), random.randint(0, h - ch)
                return x0, y0, x0 + cw, y0 + ch
        # torchvision's fallback: central crop clamped to the ratio bounds
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        x0, y0 = (w - cw) // 2, (h - ch) // 2
        return x0, y0, x0 + cw, y0 + ch

    def __call__(self, img, mask):
        img = self._jitter(img)
        box = self._crop_box(*img.size)
        img = img.crop(box).resize(_wh(self.size), Image.BILINEAR)
        mask = mask.crop(box).resize(_wh(self.size), Image.NEAREST)
        return _to_tensor_img(img), _to_tensor_mask(mask)


# labelId -> trainId of cityscapes_data.py:28-48 (255 = void)
_CITY_KEY = np.array([255, 255, 255, 255, 255, 255, 255, 255, 0, 1, 255, 255, 2, 3, 4, 255, 255, 255, 5, 255, 6, 7, 8,
                      9, 10, 11, 12, 13, 14, 15, 255, 255, 16, 17, 18], dtype=np.uint8)


class SegFolder(Dataset):
    def __init__(self, name: str, root: str, split: str, transform, file_set: Optional[List[str]] = None):
        self.name, self.root, self.split, self.transform = name, root, split, transform
        self.store = open_store(root)
        self.pairs = self._collect(file_set)
        if not self.pairs:
            raise RuntimeError(f"Dataset not found or corrupted: no {name}/{split} samples under {root}")

    def _read_lines(self, rel: str) -> List[str]:
        with self.store.open(rel) as f:
            return [ln.strip() for ln in f.read().decode().splitlines() if ln.strip()]

    def _collect(self, fs) -> List[Tuple[str, str]]:
        st = self.store
        if self.name == "voc":
            seg = "SegmentationClassAug" if self.split == "train" else "SegmentationClass"
            if not (st.isdir("images") and st.isdir(seg)):
                raise RuntimeError("Dataset not found or corrupted.")                 # voc_data.py:146-147
            if fs is None:
                fs = self._read_lines("sets/" + ("trainaug.txt" if self.split == "train" else "val.txt"))
            return [(f"images/{f}.jpg", f"{seg}/{f}.png") for f in sorted(fs)]
        if self.name == "ade20k":
            sub = "training" if self.split == "train" else "validation"
            if fs is None:
                fs = [f[:-4] for f in st.listdir(f"images/{sub}") if f.endswith(".jpg")]
            return [(f"images/{sub}/{f}.jpg", f"annotations/{sub}/{f}.png") for f in sorted(fs)]
        if self.name == "cityscapes":
            img_root, gt_root = f"leftImg8bit/{self.split}", f"gtFine/{self.split}"
            out = []
            for city in st.listdir(img_root):
                for f in st.listdir(f"{img_root}/{city}"):
                    if not f.endswith("_leftImg8bit.png"):
                        continue
                    stem = f[: -len("_leftImg8bit.png")]
                    if fs is not None and stem not in fs:
                        continue
                    out.append((f"{img_root}/{city}/{f}", f"{gt_root}/{city}/{stem}_gtFine_labelIds.png"))
            return out
        if self.name in ("coco-thing", "coco-stuff"):
            return self._collect_coco(fs)
        raise NotImplementedError(f"folder dataset '{self.name}' is not implemented")

    def _collect_coco(self, fs):
        """COCO-Stuff 'stuff' (15 coarse classes) / panoptic 'thing' (12 super-categories) masks; the category ->
        coarse-id table is derived from the annotation json exactly as coco_data.py:104-124 does, and applied as a
        256-entry lookup table on the uint8 mask (before the nearest-neighbour resize, which commutes with it)."""
        import json
        thing = self.name == "coco-thing"
        st, sp = self.store, self.split
        seg_dir = f"annotations/{sp}2017" if thing else f"annotations/stuff_annotations/stuff_{sp}2017_pixelmaps"
        js = ("annotations/panoptic_annotations/panoptic_val2017.json" if thing
              else "annotations/stuff_annotations/stuff_val2017.json")
        img_dir = f"images/{sp}2017"
        if not st.isdir(seg_dir) or not st.isdir(img_dir):
            raise RuntimeError("Dataset not found or corrupted.")                        # coco_data.py:129-132
        with st.open(js) as f:
            cats = json.loads(f.read().decode())["categories"]
        lut = np.full(256, 255, dtype=np.uint8)
        if thing:
            sup = sorted({c["supercategory"] for c in cats if c["isthing"] == 1})
            for c in cats:
                if c["isthing"] == 1 and c["id"] <= 200:
                    lut[c["id"]] = sup.index(c["supercategory"])
            lut[0] = 255                                                                  # unlabelled -> stuff -> ignored (170-178)
        else:
            sup = sorted({c["supercategory"] for c in cats} - {"other"})
            for c in cats:
                if c["id"] < 256:
                    lut[c["id"]] = 255 if c["supercategory"] == "other" else sup.index(c["supercategory"])
            lut[0] = 255                                                                  # things -> id 183 'other' -> ignored (155-160)
        self._lut = lut
        if fs is None:
            imgs = [f"{img_dir}/{f}" for f in st.listdir(img_dir)]
            msks = [f"{seg_dir}/{f}" for f in st.listdir(seg_dir)]
            return list(zip(imgs, msks))
        fs = sorted(f.replace(".jpg", "").replace(".png", "") for f in fs)
        return [(f"{img_dir}/{f}.jpg", f"{seg_dir}/{f}.png") for f in fs]

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        ip, mp = self.pairs[i]
        with self.store.open(ip) as f:
            img = Image.open(f).convert("RGB")
        with self.store.open(mp) as f:
            mask = Image.open(f)
            mask.load()
        if self.name == "cityscapes":
            ids = np.asarray(mask, dtype=np.int32)
            mask = Image.fromarray(_CITY_KEY[np.clip(ids + 1, 0, len(_CITY_KEY) - 1)])   # cityscapes_data.py:50-58
        elif self.name.startswith("coco"):
            mask = Image.fromarray(self._lut[np.asarray(mask.convert("L"), dtype=np.uint8)])
        return self.transform(img, mask)


class FolderSegDataModule:
    def __init__(self, name, data_dir, batch_size, num_workers, input_size, num_classes, train_fraction=1.0,
                 train_fs_path=None, val_fs_path=None):
        self.name, self.num_classes, self.batch_size, self.num_workers = name, num_classes, batch_size, num_workers
        tfs = read_file_set(train_fs_path) if train_fs_path else None
        vfs = read_file_set(val_fs_path) if val_fs_path else None
        self.train = SegFolder(name, data_dir, "train", TrainTransform(input_size), tfs)
        if train_fraction < 1.0:                                     # hbird/data/__init__.py:64-66
            random.shuffle(self.train.pairs)
            self.train.pairs = self.train.pairs[: int(len(self.train.pairs) * train_fraction)]
        self.val = SegFolder(name, data_dir, "val", ValTransform(input_size), vfs)

    def get_train_dataset_size(self):
        return len(self.train)

    def get_num_classes(self):
        return self.num_classes

    def _loader(self, ds):
        return DataLoader(ds, batch_size=self.batch_size, shuffle=False, num_workers=self.num_workers,
                          drop_last=False, pin_memory=True)

    def train_dataloader(self):
        return self._loader(self.train)

    def val_dataloader(self):
        return self._loader(self.val)
