"""Sliding-window tiling of large frames (BASELINE cfg-5: Cityscapes 1024 x 2048 through input_size windows).

The reference has no tiler (its loaders resize every image to input_size x input_size); this is the caller-side
format on the query AND the bank side of the hot path named as row f3 in SURVEY.md section 8:

  * bank build: `WindowedLoader` turns a loader of frames into a loader of window crops, so that
    `HbirdEvaluation._create_memory` (reference hbird_eval.py:283-369) sees ordinary input_size images;
  * evaluation: `HbirdEvaluation.evaluate(..., window=(win, stride))` runs the hot path per window and stitches the
    windows' soft predictions on the device (`hb_upsample_accumulate` + `hb_argmax_channels`).

Windows are visited in row-major order of their origins; the last row / column is flush with the frame border.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Tuple


def _axis(n: int, win: int, stride: int) -> List[int]:
    if win > n:
        raise ValueError(f"window {win} larger than the frame side {n}")
    if stride < 1:
        raise ValueError("stride must be >= 1")
    o = list(range(0, n - win + 1, stride))
    if o[-1] != n - win:
        o.append(n - win)
    return o


def window_origins(H: int, W: int, win: int, stride: int) -> List[Tuple[int, int]]:
    """(y0, x0) of every win x win window, row-major."""
    return [(y, x) for y in _axis(H, win, stride) for x in _axis(W, win, stride)]


class WindowedLoader:
    """Wraps a loader of `(x [B,3,H,W], y [B,1,H,W])` frames; yields one `(x, y)` crop batch per window."""

    def __init__(self, loader, win: int, stride: int, frame_hw: Optional[Tuple[int, int]] = None):
        self.loader, self.win, self.stride, self.frame_hw = loader, int(win), int(stride), frame_hw

    def windows_per_frame(self) -> int:
        if self.frame_hw is None:
            raise TypeError("frame_hw is needed to know the number of windows in advance")
        return len(window_origins(self.frame_hw[0], self.frame_hw[1], self.win, self.stride))

    def __len__(self) -> int:
        return len(self.loader) * self.windows_per_frame()

    def __iter__(self) -> Iterator:
        for x, y in self.loader:
            H, W = x.shape[-2:]
            for y0, x0 in window_origins(H, W, self.win, self.stride):
                yield (x[..., y0:y0 + self.win, x0:x0 + self.win].contiguous(),
                       y[..., y0:y0 + self.win, x0:x0 + self.win].contiguous())

    def rank_batches(self, keep):
        """hdist.rank_batches for window crops: ordinal = frame batch x windows-per-frame + window; a frame batch is fetched (through the
        inner loader's own rank-aware pass) only when one of its windows is kept."""
        from hbird_mi import dist as hdist
        npf = self.windows_per_frame()
        for fi, (x, y) in hdist.rank_batches(self.loader, lambda f: any(keep(f * npf + j) for j in range(npf))):
            H, W = x.shape[-2:]
            for j, (y0, x0) in enumerate(window_origins(H, W, self.win, self.stride)):
                if keep(fi * npf + j):
                    yield fi * npf + j, (x[..., y0:y0 + self.win, x0:x0 + self.win].contiguous(),
                                         y[..., y0:y0 + self.win, x0:x0 + self.win].contiguous())
