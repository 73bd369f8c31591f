"""GPU probe of the f16 storage mode: saturation (MODE.FP16_OVFL), error against the exact fp32 kernels on the default
and the stress weights (tile / slide level, per layer), and time per batch against bf16."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights

TAPS = [('block1_conv2', (147, 147, 64)), ('block2_out', (74, 74, 128)), ('block3_out', (37, 37, 256)),
        ('block4_out', (19, 19, 728))] + [(f'block{b}_out', (19, 19, 728)) for b in (5, 8, 12)] + \
       [('block13_out', (10, 10, 1024)), ('block14_sepconv1', (10, 10, 1536)), ('block14_sepconv2', (10, 10, 2048))]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tiles, sidx, _ = make_slides(max(1, n // 16), 16, seed=7)
d = torch.from_numpy(tiles).cuda()

# ---- saturation
w = synthetic_weights(1)
e16 = Engine(w, dtype='f16', max_batch=max(n, 256), max_mc=30)
big = torch.full((2, 299, 299, 3), 1.0e6, dtype=torch.float32, device='cuda')
big[1] = -3.0e5
st = e16.stage_f32(big)
print('stage_f32(1e6 / -3e5) ->', float(st[0].float().max()), float(st[1].float().min()), 'finite:', bool(torch.isfinite(st.float()).all()))
x = (torch.randn(2, 299, 299, 3, device='cuda') * 3.0e3)
a = e16.debug_activation('block1_conv2', e16.stage_f32(x), (147, 147, 64))
print('block1_conv2 on 3e3-scaled input: max', float(a.max()), 'min', float(a.min()), 'finite:', bool(torch.isfinite(a).all()),
      'n at 65504:', int((a == 65504).sum()))
a = e16.debug_activation('block2_out', e16.stage_f32(x), (74, 74, 128))
print('block2_out: max', float(a.max()), 'min', float(a.min()), 'finite:', bool(torch.isfinite(a).all()))
e16.close()

for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    e32 = Engine(w, dtype='f32', max_batch=n, max_mc=30)
    m32, s32 = e32.mc_infer(d, 30, 1234)
    sl = torch.from_numpy(sidx).cuda().long()
    S = int(sl.max()) + 1
    def smean(x):
        return torch.zeros(S, device='cuda', dtype=torch.float64).index_add_(0, sl, x.double()) / 16
    st32 = e32.stage(d[:2].contiguous())
    for dt in ('bf16', 'f16'):
        e = Engine(w, dtype=dt, max_batch=n, max_mc=30)
        m, s = e.mc_infer(d, 30, 1234)
        dm = (m32 - m).abs(); ds = (s32 - s).abs()
        print(f'hard={hard} {dt}: tile max|dmean|={float(dm.max()):.3e} max|dstd|={float(ds.max()):.3e}  '
              f'slide max|dpred|={float((smean(m32[:,1]) - smean(m[:,1])).abs().max()):.3e} '
              f'max|dunc|={float((smean(s32[:,1]) - smean(s[:,1])).abs().max()):.3e}')
        st = e.stage(d[:2].contiguous())
        for name, shp in TAPS:
            a = e32.debug_activation(name, st32, shp); b = e.debug_activation(name, st, shp)
            print(f'   {name:18s} rel rms {float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}  '
                  f'max|d| {float((a - b).abs().max()):.3e} (max|x| {float(a.abs().max()):.1f})')
        e.close()
    e32.close()

# ---- time per batch of 256, one stream
w = synthetic_weights(1)
t256 = torch.randint(0, 256, (256, 299, 299, 3), dtype=torch.uint8, device='cuda')
for dt in ('bf16', 'f16', 'bf16', 'f16'):
    e = Engine(w, dtype=dt, max_batch=256, max_mc=30)
    for _ in range(3): e.mc_infer(t256, 30, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e.mc_infer(t256, 30, 1)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
    print(f'{dt}: {dtm * 1e3:.3f} ms per batch of 256 -> {256 / dtm:.0f} tiles/s (one stream)')
    e.close()
