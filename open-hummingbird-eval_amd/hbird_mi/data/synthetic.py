"""Procedural segmentation world: C classes, each with a colour prototype; an image is a class map of random
rectangles, its RGB pixels are prototype[class] + noise.  Deterministic (own torch.Generator), no files."""
from __future__ import annotations

import torch


class _Loader:
    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class SyntheticSegDataModule:
    def __init__(self, batch_size: int = 8, input_size=224, num_classes: int = 6, n_train: int = 32,
                 n_val: int = 16, train_fraction: float = 1.0, seed: int = 0, channels: int = 3):
        self.batch_size, self.input_size, self.num_classes = batch_size, input_size, num_classes
        self.n_train = max(1, int(round(n_train * train_fraction)))
        self.n_val = n_val
        self.channels = channels
        g = torch.Generator().manual_seed(seed)
        self.proto = torch.randn((num_classes, channels), generator=g)
        self._train = self._make(self.n_train, g, with_ignore=True)
        self._val = self._make(self.n_val, g, with_ignore=True)

    def _make(self, n, g, with_ignore):
        H, W = (self.input_size, self.input_size) if isinstance(self.input_size, int) else self.input_size
        out = []
        for b0 in range(0, n, self.batch_size):
            bs = min(self.batch_size, n - b0)
            y = torch.zeros((bs, 1, H, W), dtype=torch.int64)
            for b in range(bs):
                y[b] = int(torch.randint(0, self.num_classes, (1,), generator=g))
                for _ in range(5):
                    c = int(torch.randint(0, self.num_classes, (1,), generator=g))
                    y0 = int(torch.randint(0, H, (1,), generator=g))
                    x0 = int(torch.randint(0, W, (1,), generator=g))
                    y1 = int(torch.randint(y0, H, (1,), generator=g)) + 1
                    x1 = int(torch.randint(x0, W, (1,), generator=g)) + 1
                    y[b, 0, y0:y1, x0:x1] = c
                if with_ignore:
                    y[b, 0, :2, :] = 255      # a void border like VOC's
            cls = torch.where(y[:, 0] == 255, torch.zeros_like(y[:, 0]), y[:, 0])
            x = self.proto[cls].permute(0, 3, 1, 2) + 0.3 * torch.randn((bs, self.channels, H, W), generator=g)
            out.append((x.float(), y.float() / 255.0))     # masks as ToTensor would deliver them
        return out

    def get_train_dataset_size(self):
        return self.n_train

    def get_num_classes(self):
        return self.num_classes

    def train_dataloader(self):
        return _Loader(self._train)

    def val_dataloader(self):
        return _Loader(self._val)
