"""SHA-256 of a few activations and of the MC outputs for every dtype on fixed synthetic inputs: run before and after a
kernel change that must be bit-identical (GPU only).  usage: python tools/act_hash.py [n_tiles]"""
import hashlib, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
TAPS = [('block3_out', (37, 37, 256)), ('block4_sepconv2', (37, 37, 728)), ('block4_out', (19, 19, 728)), ('block5_sepconv1', (19, 19, 728)),
        ('block5_sepconv2', (19, 19, 728)), ('block5_out', (19, 19, 728)), ('block12_out', (19, 19, 728)),
        ('block13_out', (10, 10, 1024)), ('block14_sepconv2', (10, 10, 2048))]
tiles, _, _ = make_slides(1, n, seed=5)
d = torch.from_numpy(tiles).cuda()
for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    for dt in ('f16', 'bf16'):
        e = Engine(w, dtype=dt, max_batch=max(n, 8), max_mc=8)
        st = e.stage(d)
        for name, shp in TAPS:
            a = e.debug_activation(name, st, shp).cpu().numpy()
            print(f'hard={int(hard)} {dt} {name:18s} {hashlib.sha256(a.tobytes()).hexdigest()[:16]}')
        m, s = e.mc_infer(d, 8, 1234)
        print(f'hard={int(hard)} {dt} mc_infer           {hashlib.sha256(m.cpu().numpy().tobytes() + s.cpu().numpy().tobytes()).hexdigest()[:16]}')
        e.close()
