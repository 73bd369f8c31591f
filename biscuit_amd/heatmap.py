"""Whole-slide UQ heatmap front-end (reference: results.py:216-265).

The reference builds ``sf.Heatmap(slide, model, stride_div=1)`` and masks it with the tile-level
uncertainty threshold (results.py:217-227), then walks the slide's tile grid through
``UncertaintyInterface`` and sorts the tiles into ``uq_incl`` / ``uq_excl`` (results.py:234-265).
Reading the slide itself is Slideflow/libvips work and is not rebuilt here: this front-end takes the
tiles of a grid (for example one slide's TFRecord with its ``loc_x`` / ``loc_y``) and lays the same
MC-dropout kernels' outputs out as the two grids the reference uses:

    hm.logits       [gy, gx, 2]   mean class probabilities over the MC passes
    hm.uncertainty  [gy, gx, 2]   their population std

Grid cells without a tile hold -1, the value results.py:225 also writes into masked cells.
"""
import numpy as np
import torch

MASKED = -1.0


class Heatmap:
    def __init__(self, engine, tiles, grid, grid_shape=None, mc_n=30, seed=0, batch=256, norm_fit=None):
        """tiles: uint8 [T,299,299,3] (host or device); grid: int [T,2] (gx, gy) cell of each tile."""
        grid = np.asarray(grid, dtype=np.int64).reshape(-1, 2)
        n = int(tiles.shape[0])
        if grid.shape[0] != n:
            raise ValueError(f'{n} tiles but {grid.shape[0]} grid positions')
        if n and grid.min() < 0:
            raise ValueError('negative grid position')
        if grid_shape is None:
            grid_shape = (int(grid[:, 1].max()) + 1, int(grid[:, 0].max()) + 1) if n else (0, 0)
        gy, gx = grid_shape
        if n and (grid[:, 0].max() >= gx or grid[:, 1].max() >= gy):
            raise ValueError('grid position outside grid_shape')
        self.grid = grid
        self.logits = np.full((gy, gx, 2), MASKED, dtype=np.float32)
        self.uncertainty = np.full((gy, gx, 2), MASKED, dtype=np.float32)
        dev = engine.device
        t = tiles if torch.is_tensor(tiles) else torch.from_numpy(np.ascontiguousarray(tiles))
        for s in range(0, n, batch):
            cur = t[s:s + batch].to(dev).contiguous()
            if norm_fit is not None:
                cur = engine.reinhard_fast(cur, norm_fit['target_means'], norm_fit['target_stds'])
            mean, std = engine.mc_infer(cur, mc_n, seed, tile_idx0=s)
            g = grid[s:s + batch]
            self.logits[g[:, 1], g[:, 0]] = mean.cpu().numpy()
            self.uncertainty[g[:, 1], g[:, 0]] = std.cpu().numpy()

    def mask_uncertain(self, tile_uq_thresh):
        """results.py:224-225: ``uq_mask = hm.uncertainty[:, :, 0] > thresh; hm.logits[uq_mask, :] = [-1, -1]``.
        Returns the mask."""
        uq_mask = self.uncertainty[:, :, 0] > tile_uq_thresh
        self.logits[uq_mask, :] = [MASKED, MASKED]
        return uq_mask

    def split_by_uncertainty(self, tile_uq_thresh):
        """results.py:258-265: tiles with ``uncertainty[0][0] > thresh`` go to `uq_excl`, the others to
        `uq_incl`; file names ``f"{u:.4f}-{x}-{y}.png"``.  -> (incl, excl) lists of (tile index, name)."""
        incl, excl = [], []
        for i, (x, y) in enumerate(self.grid):
            u = float(self.uncertainty[y, x, 0])
            (excl if u > tile_uq_thresh else incl).append((i, f'{u:.4f}-{x}-{y}.png'))
        return incl, excl
