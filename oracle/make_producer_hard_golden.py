"""Generate ``tests/golden/producer_hard.npz``: the fp32 CPU oracle on the STRESS weights (``synthetic_weights(1, hard=True)``:
O(1) logits, BatchNorm statistics far from the identity) for 4 slides x 16 tiles at MC = 30 -- the workload of
``tests/test_gpu_configs.py::test_hard_weights_throughput_mode_holds_tolerance``.  With it the headline parity claim -- the
f16 HIP path within 1e-3 of the reference at tile and slide level -- is held DIRECTLY against the CPU oracle on the GPU box,
not through the fp32 kernels as a go-between (round-3 review, weak item 2).  TEST INFRASTRUCTURE ONLY; the producer side
stays PARITY UNPINNED (``oracle/__init__.py``): the expected values come from this build's restatement, not from TensorFlow.

Inputs are regenerated from seeds (``make_slides(4, 16, seed=7)``, Philox seed 1234); the fixture holds the expected per-tile
mean / std, the per-slide means and a checksum of the tiles.

usage: python oracle/make_producer_hard_golden.py            (about one CPU-minute on 8 cores)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from biscuit_amd.synthetic import make_slides          # noqa: E402  (input generator only)
from biscuit_amd.weights import synthetic_weights      # noqa: E402
from oracle.xception_ref import XceptionOracle         # noqa: E402

CFG = dict(n_slides=4, tiles_per_slide=16, tile_seed=7, weight_seed=1, mc_n=30, dropout_seed=1234)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'producer_hard.npz')
    tiles, sidx, _ = make_slides(CFG['n_slides'], CFG['tiles_per_slide'], seed=CFG['tile_seed'])
    w = synthetic_weights(CFG['weight_seed'], hard=True)
    t = time.time()
    res = {}
    for tag, emu in (('f32', None), ('f16emu', 'f16')):
        mean, std = XceptionOracle(w, emulate=emu).mc_predict(tiles, CFG['mc_n'], CFG['dropout_seed'], mode='head', batch=32)
        res[f'mean_{tag}'] = mean.astype(np.float32)
        res[f'std_{tag}'] = std.astype(np.float32)
        res[f'slide_pred_{tag}'] = np.array([mean[sidx == s, 1].astype(np.float64).mean() for s in range(CFG['n_slides'])])
        res[f'slide_unc_{tag}'] = np.array([std[sidx == s, 1].astype(np.float64).mean() for s in range(CFG['n_slides'])])
        print(tag, 'done', round(time.time() - t, 1), 's', flush=True)
    np.savez_compressed(out, slide_idx=sidx, tile_checksum=np.uint64(tiles.astype(np.uint64).sum()), **res,
                        **{f'cfg_{k}': v for k, v in CFG.items()})
    print('wrote', out, os.path.getsize(out), 'bytes; pred range', res['mean_f32'][:, 1].min(), res['mean_f32'][:, 1].max(),
          'sigma range', res['std_f32'][:, 1].min(), res['std_f32'][:, 1].max())


if __name__ == '__main__':
    main()
