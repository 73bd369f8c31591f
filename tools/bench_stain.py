"""The stain normaliser alone (bq_stain_reinhard_fast: hp.py:19's `reinhard_fast`, results.py:251-252): ms per batch of 256 tiles
(HIP events, 20 repetitions after a warm-up) and the uint8 result against oracle/stain.py on 6 tiles.  python tools/bench_stain.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from biscuit_amd.engine import Engine                     # noqa: E402
from biscuit_amd.synthetic import make_tiles              # noqa: E402
from biscuit_amd.weights import synthetic_weights         # noqa: E402
from oracle import stain                                  # noqa: E402

eng = Engine(synthetic_weights(1), dtype='f16', max_batch=8, max_mc=2)
g = torch.Generator(device='cuda').manual_seed(1)
big = torch.randint(0, 256, (256, 299, 299, 3), dtype=torch.uint8, device='cuda', generator=g)
out = torch.empty_like(big)
tm, ts = [65.0, 12.0, -8.0], [14.0, 7.0, 6.0]
for _ in range(3):
    eng.reinhard_fast(big, tm, ts, out=out)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    eng.reinhard_fast(big, tm, ts, out=out)
b.record(); b.synchronize()
ms = a.elapsed_time(b) / 20
small = np.concatenate([make_tiles(3, seed=5, grain=4.0, slide_bias=[30, -40, 20]), big[:3].cpu().numpy()])
want = stain.reinhard_fast(small, np.float32(tm), np.float32(ts))
got = eng.reinhard_fast(torch.from_numpy(small).cuda(), tm, ts).cpu().numpy()
d = np.abs(got.astype(int) - want.astype(int))
st = eng.lab_stats(torch.from_numpy(small).cuda()).cpu().numpy()
L, A, B = stain.rgb_to_lab(small)
mu, sd = stain.lab_stats(L, A, B)
print(f'reinhard_fast: {ms:.4f} ms per 256 tiles ({256 * 89401 / ms / 1e6:.1f} G pixels/s; 137 MB in + out: {0.137 / ms * 1e3:.0f} GB/s); '
      f'vs oracle on 6 tiles: max |d| {d.max()}, differing bytes {int((d != 0).sum())} of {d.size}; '
      f'stats max |d| {max(np.abs(st[:, :3] - mu).max(), np.abs(st[:, 3:] - sd).max()):.2e}')
