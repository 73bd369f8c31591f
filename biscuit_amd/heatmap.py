"""Whole-slide UQ heatmap front-end (reference: results.py:216-265).

The reference builds ``sf.Heatmap(slide, model, stride_div=1)`` and masks it with the tile-level
uncertainty threshold (results.py:217-227), then walks the slide's tile grid through
``UncertaintyInterface`` and sorts the tiles into ``uq_incl`` / ``uq_excl`` (results.py:234-265).
This front-end takes the tiles of a grid -- one slide's TFRecord with its ``loc_x`` / ``loc_y``, a region in memory
(``from_region``), or a pyramidal TIFF / SVS slide file through this build's own reader (``from_slide``, ``biscuit_amd/wsi.py``,
round 6) -- and lays the same MC-dropout kernels' outputs out as the two grids the reference uses:

    hm.logits       [gy, gx, 2]   mean class probabilities over the MC passes
    hm.uncertainty  [gy, gx, 2]   their population std

Grid cells without a tile hold -1, the value results.py:225 also writes into masked cells.
"""
import numpy as np
import torch

MASKED = -1.0


def tile_grid(region, tile_px=299, stride_div=1):
    """The tile grid ``sf.Heatmap(slide, model, stride_div=...)`` walks (results.py:217), over a slide region that is
    already in memory: tiles of ``tile_px`` at stride ``tile_px // stride_div``, row-major, border remainders dropped.
    region: uint8 [H, W, 3] (numpy or torch, host or device).  Returns (tiles [T, tile_px, tile_px, 3] -- a torch
    tensor on the region's device -- and grid int64 [T, 2] of (gx, gy) cells)."""
    r = region if torch.is_tensor(region) else torch.from_numpy(np.ascontiguousarray(region))
    if r.dim() != 3 or r.shape[2] != 3 or r.dtype != torch.uint8:
        raise ValueError('region must be uint8 [H, W, 3]')
    if stride_div < 1 or tile_px % stride_div:
        raise ValueError('stride_div must divide tile_px')
    stride = tile_px // stride_div
    gy = (r.shape[0] - tile_px) // stride + 1 if r.shape[0] >= tile_px else 0
    gx = (r.shape[1] - tile_px) // stride + 1 if r.shape[1] >= tile_px else 0
    if gy == 0 or gx == 0:
        return r.new_zeros((0, tile_px, tile_px, 3)), np.zeros((0, 2), np.int64)
    # [gy, gx, 3, tile_px, tile_px] view -> NHWC tiles
    t = r.unfold(0, tile_px, stride).unfold(1, tile_px, stride)
    tiles = t.permute(0, 1, 3, 4, 2).reshape(gy * gx, tile_px, tile_px, 3).contiguous()
    ys, xs = np.divmod(np.arange(gy * gx, dtype=np.int64), gx)
    return tiles, np.stack([xs, ys], 1)


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

    @classmethod
    def from_region(cls, engine, region, tile_px=299, stride_div=1, **kw):
        """Heatmap of a slide region in memory: the stride-``tile_px // stride_div`` grid of ``tile_grid``."""
        tiles, grid = tile_grid(region, tile_px, stride_div)
        if len(grid) == 0:
            raise ValueError(f'region {tuple(region.shape[:2])} holds no {tile_px} x {tile_px} tile')
        shape = (int(grid[:, 1].max()) + 1, int(grid[:, 0].max()) + 1)      # (gy, gx) of tile_grid's clamped grid
        return cls(engine, tiles, grid, grid_shape=shape, **kw)

    @classmethod
    def from_slide(cls, engine, path, tile_px=299, tile_um=302, stride_div=1, mpp=None, **kw):
        """``sf.Heatmap(slide, model, stride_div=1)`` (results.py:217) for a pyramidal TIFF / SVS slide file: the tile grid of
        ``wsi.WSI(path, tile_px, tile_um, stride_div)`` through the MC-dropout kernels.  (The reader is this build's own --
        ``biscuit_amd/wsi.py`` says what it reads and what about it is unpinned.)"""
        from .wsi import WSI
        w = WSI(path, tile_px=tile_px, tile_um=tile_um, stride_div=stride_div, mpp=mpp)
        try:
            tiles, grid = w.tiles()
            if len(grid) == 0:
                raise ValueError(f'{path}: the slide holds no {tile_um} um tile')
            return cls(engine, tiles, grid, grid_shape=(w.grid_h, w.grid_w), **kw)
        finally:
            w.close()

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
