"""bf16 kernels against the exact fp32 kernels (which match the CPU oracle to 1e-7) on the default and the
stress weights: tile-level max|d| of mean / std at MC=30 and the per-layer relative RMS error.  GPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights

TAPS = [('block1_conv2', (147, 147, 64)), ('block2_out', (74, 74, 128)), ('block3_out', (37, 37, 256)),
        ('block4_out', (19, 19, 728))] + [(f'block{b}_out', (19, 19, 728)) for b in range(5, 13)] + \
       [('block13_out', (10, 10, 1024)), ('block14_sepconv1', (10, 10, 1536)), ('block14_sepconv2', (10, 10, 2048))]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tiles, sidx, _ = make_slides(max(1, n // 16), 16, seed=7)
d = torch.from_numpy(tiles).cuda()
for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    e32 = Engine(w, dtype='f32', max_batch=n, max_mc=30)
    e16 = Engine(w, dtype='bf16', max_batch=n, max_mc=30)
    m32, s32 = e32.mc_infer(d, 30, 1234)
    m16, s16 = e16.mc_infer(d, 30, 1234)
    dm = (m32 - m16).abs(); ds = (s32 - s16).abs()
    print(f'hard={hard}: n={len(tiles)} pred range [{float(m32[:,1].min()):.3f},{float(m32[:,1].max()):.3f}] '
          f'std mean {float(s32[:,1].mean()):.4f}  max|dmean|={float(dm.max()):.3e} (mean {float(dm.mean()):.2e}) '
          f'max|dstd|={float(ds.max()):.3e}')
    sl = torch.from_numpy(sidx).cuda().long()
    S = int(sl.max()) + 1
    def smean(x):
        return torch.zeros(S, device='cuda', dtype=torch.float64).index_add_(0, sl, x.double()) / 16
    print(f'   slide-level max|dpred|={float((smean(m32[:,1]) - smean(m16[:,1])).abs().max()):.3e} '
          f'max|dunc|={float((smean(s32[:,1]) - smean(s16[:,1])).abs().max()):.3e}')
    f32 = e32.backbone(e32.stage(d[:8].contiguous())); f16 = e16.backbone(e16.stage(d[:8].contiguous()))
    print(f'   features rel rms {float((f32 - f16).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()):.3e}')
    st32 = e32.stage(d[:2].contiguous()); st16 = e16.stage(d[:2].contiguous())
    for name, shp in TAPS:
        a = e32.debug_activation(name, st32, shp); b = e16.debug_activation(name, st16, shp)
        print(f'   {name:18s} rel rms {float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}  max|d| {float((a - b).abs().max()):.3e} (max|x| {float(a.abs().max()):.1f})')
    e32.close(); e16.close()
