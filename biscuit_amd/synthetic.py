"""Seeded synthetic slides of 299x299x3 uint8 tiles (there is no network for real
TCGA TFRecords; tile geometry from ``configure.py:118-124`` / ``biscuit/hp.py:5``).

Tiles are low-frequency colour texture plus noise rather than i.i.d. noise, with a
per-slide stain-like colour bias, so per-image standardisation does not collapse
every tile to the same statistics and slide means differ.
"""
import numpy as np

TILE_PX = 299


def make_tiles(n_tiles, seed, slide_bias=None, px=TILE_PX, grain=18.0):
    """[n_tiles, px, px, 3] uint8 (NHWC, the TFRecord decode layout).  ``grain``: standard deviation of the per-pixel noise -- 18
    (the default of every fixture) makes a tile nearly incompressible (a 226 KB PNG: Sub / Up rows, the worst case for inflate), 4 a
    photo-like one (smooth texture + sensor grain: a 155 KB PNG, Paeth / Average rows -- the size of a real H&E tile)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(px, dtype=np.float32), np.arange(px, dtype=np.float32),
                         indexing='ij')
    out = np.empty((n_tiles, px, px, 3), np.uint8)
    bias = np.zeros(3, np.float32) if slide_bias is None else np.asarray(slide_bias, np.float32)
    for t in range(n_tiles):
        img = np.zeros((px, px, 3), np.float32)
        for _ in range(4):
            fx, fy = rng.uniform(0.005, 0.08, 2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(10, 45, 3).astype(np.float32)
            img += np.cos(fx * xx + fy * yy + ph)[:, :, None] * amp
        img += rng.normal(0, grain, (px, px, 3)).astype(np.float32)
        img += 128 + bias + rng.normal(0, 12, 3).astype(np.float32)
        out[t] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def make_slides(n_slides, tiles_per_slide, seed=0):
    """Returns (tiles uint8 [S*T,299,299,3], slide_idx int32 [S*T], y_true int [S])."""
    rng = np.random.default_rng(seed)
    tiles, idx = [], []
    for s in range(n_slides):
        bias = rng.normal(0, 25, 3)
        tiles.append(make_tiles(tiles_per_slide, seed * 100003 + s + 1, bias))
        idx.append(np.full(tiles_per_slide, s, np.int32))
    y_true = (np.arange(n_slides) % 2).astype(np.int64)
    return np.concatenate(tiles), np.concatenate(idx), y_true
