"""Sliding-window tiler (BASELINE cfg-5): host logic and the oracle's stitching.  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from hbird_mi import tiling


@pytest.mark.parametrize("H,W,win,stride", [(1024, 2048, 518, 350), (64, 64, 64, 64), (70, 100, 32, 32), (96, 64, 32, 16),
                                            (33, 47, 32, 7)])
def test_window_origins_cover_the_frame(H, W, win, stride):
    o = tiling.window_origins(H, W, win, stride)
    assert o == oracle.window_origins(H, W, win, stride)
    assert o == sorted(o)                                   # row-major
    cover = np.zeros((H, W), dtype=np.int32)
    for y0, x0 in o:
        assert 0 <= y0 <= H - win and 0 <= x0 <= W - win
        cover[y0:y0 + win, x0:x0 + win] += 1
    assert cover.min() >= 1
    assert (0, 0) in o and (H - win, W - win) in o


def test_window_origins_errors():
    with pytest.raises(ValueError):
        tiling.window_origins(30, 64, 32, 8)
    with pytest.raises(ValueError):
        tiling.window_origins(64, 64, 32, 0)


def test_windowed_loader_yields_crops_in_origin_order():
    frames = [(torch.arange(2 * 3 * 40 * 72, dtype=torch.float32).reshape(2, 3, 40, 72) + 1000 * i,
               torch.rand(2, 1, 40, 72)) for i in range(3)]
    wl = tiling.WindowedLoader(frames, 32, 24, frame_hw=(40, 72))
    origins = tiling.window_origins(40, 72, 32, 24)
    assert wl.windows_per_frame() == len(origins) == 2 * 3
    assert len(wl) == 3 * len(origins)
    out = list(wl)
    assert len(out) == len(wl)
    for i, (x, y) in enumerate(out):
        fx, fy = frames[i // len(origins)]
        y0, x0 = origins[i % len(origins)]
        assert x.shape == (2, 3, 32, 32) and y.shape == (2, 1, 32, 32) and x.is_contiguous()
        assert torch.equal(x, fx[..., y0:y0 + 32, x0:x0 + 32]) and torch.equal(y, fy[..., y0:y0 + 32, x0:x0 + 32])
    with pytest.raises(TypeError):
        len(tiling.WindowedLoader(frames, 32, 24))


def test_oracle_stitching_reduces_to_the_per_image_path_for_one_window():
    rng = np.random.default_rng(0)
    S, C, H = 4, 5, 32
    lh = rng.random((2, S * S, C), dtype=np.float32)
    cm, acc = oracle.sliding_window_argmax([lh], [(0, 0)], S, H, H, H)
    assert np.array_equal(cm, oracle.upsample_argmax(lh, S, H, H))
    up = oracle.upsample_bilinear(lh.reshape(2, S, S, C).transpose(0, 3, 1, 2), H, H)
    assert np.array_equal(acc, up)


def test_oracle_stitching_sums_overlaps():
    rng = np.random.default_rng(1)
    S, C, win, H, W = 4, 3, 16, 16, 24
    origins = oracle.window_origins(H, W, win, 8)
    assert origins == [(0, 0), (0, 8)]
    lhs = [rng.random((1, S * S, C), dtype=np.float32) for _ in origins]
    cm, acc = oracle.sliding_window_argmax(lhs, origins, S, win, H, W)
    ups = [oracle.upsample_bilinear(l.reshape(1, S, S, C).transpose(0, 3, 1, 2), win, win) for l in lhs]
    assert np.array_equal(acc[..., :8], ups[0][..., :8])
    assert np.array_equal(acc[..., 8:16], ups[0][..., 8:] + ups[1][..., :8])
    assert np.array_equal(acc[..., 16:], ups[1][..., 8:])
    assert np.array_equal(cm[:, 0], acc.argmax(1))


def test_cli_accepts_frame_flags():
    import eval as cli
    a = cli.build_parser().parse_args("--dataset-name synthetic --data-dir x --d-model 8 --patch-size 4 --input-size 32 "
                                      "--frame-size 32 64 --window-stride 16".split())
    assert a.frame_size == [32, 64] and a.window_stride == 16


def test_synthetic_module_delivers_rectangular_frames():
    from hbird_mi.data import get_dataset
    dm, ign = get_dataset("synthetic", "", 4, 0, (32, 48))
    x, y = next(iter(dm.val_dataloader()))
    assert x.shape[-2:] == (32, 48) and y.shape[-2:] == (32, 48) and ign == 255
