import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from biscuit_amd.engine import Engine
from biscuit_amd.weights import synthetic_weights
from oracle.xception_ref import XceptionOracle
w = synthetic_weights(1)
eng = Engine(w, dtype='f32', max_batch=64, max_mc=50)
orc = XceptionOracle(w)
feat = np.abs(np.random.default_rng(3).normal(0.8, 0.5, (37, 2048))).astype(np.float32)
for mc in (1, 5, 30):
    m, s = eng.mc_head(torch.from_numpy(feat).cuda(), mc, 1234, tile_idx0=11)
    rm, rs = orc.mc_from_features(feat, mc, 1234, tile_index0=11)
    m, s = m.cpu().numpy(), s.cpu().numpy()
    print('mc', mc, 'dmean', np.abs(m - rm).max(), 'dstd', np.abs(s - rs).max())
    i = int(np.abs(s - rs).max(axis=1).argmax())
    print('  row', i, 'dev mean', m[i], 'ref mean', rm[i], 'dev std', s[i], 'ref std', rs[i])
