"""CPU estimate of the storage-type error: the oracle with bf16 / f16 rounding points against the fp32 oracle on the
default and the stress weights (tile-level max|d| of MC mean / std, slide means of 16 tiles, activation ranges)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from oracle.xception_ref import XceptionOracle, standardize
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mc = 30
tiles, sidx, _ = make_slides(max(1, n // 16), 16, seed=7)
for hard in (False, True):
    w = synthetic_weights(1, hard=hard)
    ref = XceptionOracle(w)
    taps = {}
    x = standardize(tiles)
    f32 = torch.cat([ref.backbone(x[i:i + 16], taps if i == 0 else None) for i in range(0, n, 16)])
    m32, s32 = ref.mc_from_features(f32, mc, 1234)
    print(f'hard={hard}: max|act| over taps = {max(float(t.abs().max()) for t in taps.values()):.1f}; '
          f'min nonzero |act| quantile 1e-3: {min(float(t.abs()[t != 0].quantile(1e-3)) if t.numel() < 16e6 else 0 for t in taps.values()):.2e}')
    for emu in ('bf16', 'f16'):
        o = XceptionOracle(w, emulate=emu)
        f = torch.cat([o.backbone(x[i:i + 16]) for i in range(0, n, 16)])
        m, s = o.mc_from_features(f, mc, 1234)
        dm, ds = np.abs(m - m32).max(), np.abs(s - s32).max()
        S = n // 16
        sm = np.abs(m[:, 1].reshape(S, 16).mean(1) - m32[:, 1].reshape(S, 16).mean(1)).max()
        ss = np.abs(s[:, 1].reshape(S, 16).mean(1) - s32[:, 1].reshape(S, 16).mean(1)).max()
        print(f'  {emu}: tile max|dmean| {dm:.3e} max|dstd| {ds:.3e}   slide max|dpred| {sm:.3e} max|dunc| {ss:.3e}  '
              f'feat rel rms {float((f - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()):.3e}')
