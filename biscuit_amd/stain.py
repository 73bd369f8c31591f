"""Stain normalisation in front of the staging kernel (hp.py:19 ``normalizer='reinhard_fast'``).

Mirrors the object the reference calls at results.py:251-252
(``interface.wsi_normalizer.rgb_to_rgb(image)``): a fitted normaliser with ``rgb_to_rgb``, ``fit`` and
``get_fit``; the arithmetic runs in the HIP kernel behind ``bq_stain_reinhard_fast``.  The target
statistics are the ``norm_fit`` block of a Slideflow model's ``params.json`` -- they are read from
there (``from_params``) or fitted to a target image, never hard-coded.
"""
import json

import numpy as np
import torch

TILE_PX = 299


class ReinhardFast:
    method = 'reinhard_fast'

    def __init__(self, engine, target_means=None, target_stds=None):
        self.engine = engine
        self.target_means = None if target_means is None else np.asarray(target_means, np.float32).reshape(3)
        self.target_stds = None if target_stds is None else np.asarray(target_stds, np.float32).reshape(3)

    @classmethod
    def from_params(cls, engine, params):
        """``params``: dict or path of a Slideflow params.json holding ``norm_fit``."""
        if not isinstance(params, dict):
            with open(params) as f:
                params = json.load(f)
        fit = params.get('norm_fit')
        if not fit or 'target_means' not in fit or 'target_stds' not in fit:
            raise ValueError("params.json has no norm_fit with target_means / target_stds")
        return cls(engine, fit['target_means'], fit['target_stds'])

    def _as_batch(self, image):
        t = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(image))
        if t.dtype != torch.uint8:
            raise TypeError('stain normalisation takes uint8 RGB')
        single = t.dim() == 3
        t = t.unsqueeze(0) if single else t
        if tuple(t.shape[1:]) != (TILE_PX, TILE_PX, 3):
            raise ValueError(f'expected [n,{TILE_PX},{TILE_PX},3] uint8 tiles, got {tuple(t.shape)}')
        return t.to(self.engine.device).contiguous(), single

    def fit(self, target):
        """Fit to one target tile [299,299,3] uint8: stores its CIE-LAB channel means / stds."""
        t, _ = self._as_batch(target)
        st = self.engine.lab_stats(t[:1]).cpu().numpy()[0]
        self.target_means, self.target_stds = st[:3].copy(), st[3:].copy()
        return self

    def get_fit(self):
        return {'target_means': self.target_means.tolist(), 'target_stds': self.target_stds.tolist()}

    def rgb_to_rgb(self, image):
        if self.target_means is None:
            raise RuntimeError('normaliser is not fitted (fit() or from_params())')
        t, single = self._as_batch(image)
        out = self.engine.reinhard_fast(t, self.target_means, self.target_stds)
        return out[0] if single else out
