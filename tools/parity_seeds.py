"""The headline mode's tolerance over several draws of the stress weights (GPU box): f16 and bf16 kernels against the exact
fp32 kernels, MC = 30, per weight seed: tile max|d mean|, max|d std|, slide max|d pred|, max|d unc|.
usage: python tools/parity_seeds.py [--tiles N] [seed ...]      (round 4: 256 tiles per draw = 4 slides x 64; round 3 ran 64)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from biscuit_amd.engine import Engine                     # noqa: E402
from biscuit_amd.synthetic import make_slides             # noqa: E402
from biscuit_amd.weights import synthetic_weights         # noqa: E402

args = sys.argv[1:]
NT = 256
if '--tiles' in args:
    i = args.index('--tiles')
    NT = int(args[i + 1])
    del args[i:i + 2]
TPS = NT // 4                                              # tiles per slide
seeds = [int(a) for a in args] or list(range(1, 9))
print(f'{NT} tiles per draw (4 slides x {TPS}), MC = 30, seeds {seeds}')
worst = {'f16': np.zeros(4), 'bf16': np.zeros(4)}
for ws in seeds:
    w = synthetic_weights(ws, hard=True)
    tiles, sidx, _ = make_slides(4, TPS, seed=100 + ws)
    d = torch.from_numpy(tiles).cuda()
    sl = torch.from_numpy(sidx).cuda().long()
    eng = {t: Engine(w, dtype=t, max_batch=NT, max_mc=30) for t in ('f32', 'f16', 'bf16')}
    m32, s32 = eng['f32'].mc_infer(d, 30, 1234)

    def smean(x):
        return torch.zeros(4, device='cuda', dtype=torch.float64).index_add_(0, sl, x.double()) / TPS
    for t in ('f16', 'bf16'):
        m, s = eng[t].mc_infer(d, 30, 1234)
        r = np.array([float((m32 - m).abs().max()), float((s32 - s).abs().max()),
                      float((smean(m32[:, 1]) - smean(m[:, 1])).abs().max()), float((smean(s32[:, 1]) - smean(s[:, 1])).abs().max())])
        worst[t] = np.maximum(worst[t], r)
        print(f'weights seed {ws} {t}: tile dmean {r[0]:.2e} dstd {r[1]:.2e}  slide dpred {r[2]:.2e} dunc {r[3]:.2e}'
              f'   (p range {float(m32[:, 1].min()):.2f}..{float(m32[:, 1].max()):.2f}, sigma max {float(s32.max()):.3f})', flush=True)
    for e in eng.values():
        e.close()
for t in worst:
    print(f'worst over {len(seeds)} seeds, {t}: tile dmean {worst[t][0]:.2e} dstd {worst[t][1]:.2e}  slide dpred {worst[t][2]:.2e} dunc {worst[t][3]:.2e}')
