"""Same-box A/B of the MC head's dense kernels (bq_set_option head_variant 0 / 1 / 2): bit equality and time per launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd.engine import Engine
from biscuit_amd.weights import synthetic_weights
eng = Engine(synthetic_weights(1), dtype='f16', max_batch=256, max_mc=30)
feat = torch.rand((256, 2048), device='cuda') * 2
ref = None
for rep in range(3):
    for v in (0, 1, 2):
        eng.set_option('head_variant', v)
        m, s = eng.mc_head(feat, 30, 7, tile_idx0=11)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            eng.mc_head(feat, 30, 7, tile_idx0=11)
        b.record(); b.synchronize()
        if ref is None:
            ref = (m.clone(), s.clone())
        same = torch.equal(m, ref[0]) and torch.equal(s, ref[1])
        print(f'rep {rep} head_variant {v}: {a.elapsed_time(b) / 50:.4f} ms per head (dense0 + dense1 + final), bit-identical to variant 0: {same}', flush=True)
# a ragged row count and one tile (the B = 1 latency path)
for n in (1, 37):
    f = feat[:n].contiguous()
    outs = []
    for v in (0, 1, 2):
        eng.set_option('head_variant', v)
        outs.append(eng.mc_head(f, 30, 7, tile_idx0=3))
    print('n', n, 'equal', all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs))
