"""Per-layer distance of the 16-bit kernels from the oracle that rounds at the same points, in units of the storage
type's ulp at the layer's largest magnitude (max-abs) and of the relative ulp (rms): the figures behind the bounds of
tests/test_gpu_parity.py::test_every_layer_against_oracle.  GPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights
from oracle.xception_ref import XceptionOracle, standardize
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from test_gpu_parity import TAPS, ULP

for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    tiles, _, _ = make_slides(3, 2, seed=0)
    t2 = tiles[:2]
    for dt in ('bf16', 'f16'):
        taps = {}
        fr = XceptionOracle(w, emulate=dt).backbone(standardize(t2), taps).numpy()
        e = Engine(w, dtype=dt, max_batch=8, max_mc=8)
        st = e.stage(torch.from_numpy(t2).cuda())
        worst_a = worst_r = 0
        for name, shp in TAPS:
            got = e.debug_activation(name, st, shp).cpu().numpy()
            ref = taps[name].permute(0, 2, 3, 1).numpy()
            d = np.abs(got - ref)
            a = d.max() / (ULP[dt] * np.abs(ref).max())
            r = np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()) / ULP[dt]
            frac = (d > 0).mean()
            worst_a, worst_r = max(worst_a, a), max(worst_r, r)
            print(f'hard={hard} {dt} {name:18s} max|d| {a:6.2f} ulp@max  rms {r:6.3f} rel-ulp  differing {frac:.4f}')
        f = e.backbone(st).cpu().numpy()
        print(f'hard={hard} {dt} WORST max {worst_a:.2f} rms {worst_r:.3f}; features max|d| {np.abs(f - fr).max() / (ULP[dt] * np.abs(fr).max()):.3f} ulp@max')
        e.close()
