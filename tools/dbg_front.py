"""The fused front kernel (uint8 -> block1_conv2) against the oracle and against the three kernels it replaces (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights
from oracle.xception_ref import XceptionOracle, standardize

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    tiles = make_tiles(n, seed=3)
    tiles[1] = (tiles[1] * 0.2 + 90).astype(np.uint8)          # a low-contrast tile
    for dt in ('f16', 'bf16'):
        taps = {}
        XceptionOracle(w, emulate=dt).backbone(standardize(tiles), taps)
        e = Engine(w, dtype=dt, max_batch=max(n, 8), max_mc=8)
        d = torch.from_numpy(tiles).cuda()
        st = e.stage(d)
        for name, shp in (('block1_conv2', (147, 147, 64)), ('block2_out', (74, 74, 128)), ('block3_out', (37, 37, 256))):
            ref = taps[name].permute(0, 2, 3, 1).numpy()
            old = e.debug_activation(name, st, shp).cpu().numpy()
            new = e.debug_activation_u8(name, d, shp).cpu().numpy()
            ulp = (2.0 ** -8 if dt == 'bf16' else 2.0 ** -11) * np.abs(ref).max()
            dn, do = np.abs(new - ref), np.abs(old - ref)
            print(f'hard={int(hard)} {dt} {name}: new vs oracle {dn.max() / ulp:.2f} ulps (old {do.max() / ulp:.2f}), new vs old {np.abs(new - old).max() / ulp:.2f} ulps, '
                  f'differing {float((new != old).mean()):.2e}')
            if dn.max() > 4 * ulp:
                bad = dn > 4 * ulp
                print('  bad frac', bad.mean(), 'rows', np.flatnonzero(bad.any(axis=(0, 2, 3)))[:20], 'cols', np.flatnonzero(bad.any(axis=(0, 1, 3)))[:20],
                      'ch', np.flatnonzero(bad.any(axis=(0, 1, 2)))[:20], 'img', np.flatnonzero(bad.any(axis=(1, 2, 3))))
        e.close()
