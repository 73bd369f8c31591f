"""Generate the producer-side golden fixture for BASELINE.json config 1 ("16 synthetic
slides x 64 random 299x299x3 tiles, Xception, MC=5, CPU reference path") with the fp32
CPU oracle.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED on this side (see
``oracle/__init__.py``): the expected values come from this build's restatement, not from
TensorFlow/Slideflow, which cannot be installed here.

Inputs are not stored: tiles and weights are regenerated from seeds
(``biscuit_amd.synthetic.make_slides(16, 64, seed=0)``, ``synthetic_weights(1)``, Philox
seed 1234); the fixture holds the expected per-tile mean/std and per-slide means.

usage: python oracle/make_producer_golden.py [tag ...]     (about 2 CPU-minutes per tag on 8 cores)
tags: f32 (the parity oracle), bf16emu, f16emu (the oracle with the 16-bit paths' rounding points).  Without arguments
all three are generated; with arguments only the named ones are recomputed and the others kept from the existing file.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from biscuit_amd.synthetic import make_slides          # noqa: E402  (input generator only)
from biscuit_amd.weights import synthetic_weights      # noqa: E402
from oracle.xception_ref import XceptionOracle         # noqa: E402

CFG1 = dict(n_slides=16, tiles_per_slide=64, tile_seed=0, weight_seed=1, mc_n=5, dropout_seed=1234)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'producer_cfg1.npz')
    tiles, sidx, y_true = make_slides(CFG1['n_slides'], CFG1['tiles_per_slide'], seed=CFG1['tile_seed'])
    w = synthetic_weights(CFG1['weight_seed'])
    t = time.time()
    res = {}
    modes = {'f32': None, 'bf16emu': 'bf16', 'f16emu': 'f16'}
    want = sys.argv[1:] or list(modes)
    if os.path.exists(out):                      # keep what is not being regenerated
        old = np.load(out)
        assert np.uint64(tiles.astype(np.uint64).sum()) == old['tile_checksum']
        res = {k: old[k] for k in old.files if k.split('_')[-1] in modes and k.split('_')[-1] not in want}
    for tag in want:
        orc = XceptionOracle(w, emulate=modes[tag])
        mean, std = orc.mc_predict(tiles, CFG1['mc_n'], CFG1['dropout_seed'], mode='head', batch=32)
        res[f'mean_{tag}'] = mean.astype(np.float32)
        res[f'std_{tag}'] = std.astype(np.float32)
        # slide-level reduce in float64, the dtype pandas gives the reference (threshold.py:191-192)
        res[f'slide_pred_{tag}'] = np.array([mean[sidx == s, 1].astype(np.float64).mean()
                                             for s in range(CFG1['n_slides'])])
        res[f'slide_unc_{tag}'] = np.array([std[sidx == s, 1].astype(np.float64).mean()
                                            for s in range(CFG1['n_slides'])])
        print(tag, 'done', time.time() - t, flush=True)
    np.savez_compressed(out, slide_idx=sidx, y_true=y_true, tile_checksum=np.uint64(tiles.astype(np.uint64).sum()),
                        **res, **{f'cfg_{k}': v for k, v in CFG1.items()})
    print('wrote', out, os.path.getsize(out))


if __name__ == '__main__':
    main()
