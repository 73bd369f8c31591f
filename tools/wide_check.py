"""bf16 wide-kernel layers against the fp32 kernels, layer by layer (GPU only): finite? relative rms?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights
TAPS = [('block2_out', (74, 74, 128)), ('block3_sepconv1', (74, 74, 256)), ('block3_sepconv2', (74, 74, 256)), ('block3_out', (37, 37, 256)),
        ('block4_sepconv1', (37, 37, 728)), ('block4_sepconv2', (37, 37, 728)), ('block4_out', (19, 19, 728)), ('block5_sepconv1', (19, 19, 728)), ('block5_sepconv2', (19, 19, 728)), ('block5_out', (19, 19, 728)),
        ('block12_out', (19, 19, 728)), ('block13_out', (10, 10, 1024))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = synthetic_weights(1)
e32 = Engine(w, dtype='f32', max_batch=n, max_mc=4); e16 = Engine(w, dtype='bf16', max_batch=n, max_mc=4)
d = torch.from_numpy(make_tiles(n, seed=3)).cuda()
s32, s16 = e32.stage(d), e16.stage(d)
for name, shp in TAPS:
    a = e32.debug_activation(name, s32, shp); b = e16.debug_activation(name, s16, shp)
    bad = ~torch.isfinite(b)
    msg = f'{name:16s} nonfinite {int(bad.sum())}'
    if bad.any():
        idx = bad.nonzero()[:4].tolist()
        msg += f' first at (img,y,x,c) {idx}'
        b = torch.where(bad, torch.zeros_like(b), b)
    err = (a - b).abs()
    msg += f'  rel rms {float(err.pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}  max|d| {float(err.max()):.3e}'
    worst = (err == err.max()).nonzero()[0].tolist()
    print(msg, 'worst at', worst)
