"""Where the streaming kernel's outputs differ from the oracle (GPU only): error by row, column and channel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights
from oracle.xception_ref import XceptionOracle, standardize

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = synthetic_weights(1)
tiles = make_tiles(n, seed=3)
for dt in ('f16', 'bf16'):
    taps = {}
    XceptionOracle(w, emulate=dt).backbone(standardize(tiles), taps)
    e = Engine(w, dtype=dt, max_batch=max(n, 8), max_mc=8)
    st = e.stage(torch.from_numpy(tiles).cuda())
    for name, shp in (('block1_conv2', (147, 147, 64)), ('block2_sepconv1', (147, 147, 128)), ('block2_sepconv2', (147, 147, 128)), ('block2_out', (74, 74, 128))):
        got = e.debug_activation(name, st, shp).cpu().numpy()
        ref = taps[name].permute(0, 2, 3, 1).numpy()
        d = np.abs(got - ref)
        ulp = (2.0 ** -8 if dt == 'bf16' else 2.0 ** -11) * np.abs(ref).max()
        bad = d > 4 * ulp
        print(dt, name, 'max ulps', d.max() / ulp, 'bad frac', bad.mean())
        if bad.any():
            print('  bad by image', bad.reshape(n, -1).mean(1))
            print('  bad rows   ', np.flatnonzero(bad.any(axis=(0, 2, 3)))[:40])
            print('  bad cols   ', np.flatnonzero(bad.any(axis=(0, 1, 3)))[:40])
            print('  bad chans  ', np.flatnonzero(bad.any(axis=(0, 1, 2)))[:40])
            i = np.unravel_index(np.argmax(d), d.shape)
            print('  worst', i, got[i], ref[i])
    e.close()
