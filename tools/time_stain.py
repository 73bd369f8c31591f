import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd.engine import Engine
from biscuit_amd.weights import synthetic_weights
eng = Engine(synthetic_weights(1), dtype='bf16', max_batch=256, max_mc=4)
t = torch.randint(0, 256, (256, 299, 299, 3), dtype=torch.uint8, device='cuda')
out = torch.empty_like(t)
tm, ts = [60., 10., -5.], [15., 8., 6.]
for fn, name in ((lambda: eng.reinhard_fast(t, tm, ts, out=out), 'reinhard_fast'), (lambda: eng.lab_stats(t), 'lab_stats'), (lambda: eng.stage(t), 'stage')):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'{name}: {dt*1e3:.3f} ms per 256 tiles')
