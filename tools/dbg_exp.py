"""Developer probe (GPU): where do forced activation exponents change a stored value?  Every tap of the f16 path with exponents k
against the same path without, at true scale."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights, tensor_plan, tensor_taps

k = int(sys.argv[1]) if len(sys.argv) > 1 else -2
w = synthetic_weights(3, hard=True)
tiles, _, _ = make_slides(1, 2, seed=41)
d = torch.from_numpy(tiles).cuda()
e2 = Engine(w, dtype='f16', max_batch=4, max_mc=4)
e3 = Engine(w, dtype='f16', max_batch=4, max_mc=4, act_exp={t: k for _, _, t in tensor_plan()})
order = ['block1_conv2'] + [l for l, _, _ in tensor_plan() if l not in ('block1_conv1', 'block1_conv2') and not l.endswith('_sepconv3')]
for b in (2, 3, 4, 13):
    order.insert(order.index(f'block{b}_res') + 1, f'block{b}_out')
for b in range(5, 13):
    order.insert(order.index(f'block{b}_sepconv2') + 1, f'block{b}_out')
for tap in order:
    shp = Engine.TAP_SHAPES[tap]
    try:
        a = e2.debug_activation_u8(tap, d, shp); b = e3.debug_activation_u8(tap, d, shp)
    except Exception as ex:
        print(tap, 'ERR', ex); continue
    diff = (a - b).abs()
    nz = int((diff > 0).sum())
    print(f'{tap:18s} max|a| {float(a.abs().max()):9.3f}  differing {nz:8d} of {a.numel():9d}  max diff {float(diff.max()):.3e}  smallest |a| among differing '
          f'{float(a.abs()[diff > 0].min()) if nz else 0:.3e}  largest {float(a.abs()[diff > 0].max()) if nz else 0:.3e}', flush=True)
f2, f3 = e2.backbone_u8(d), e3.backbone_u8(d)
print('features max diff', float((f2 - f3).abs().max()), 'of', float(f2.abs().max()))
