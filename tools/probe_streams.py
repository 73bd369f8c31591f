import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from biscuit_amd.engine import Engine
from biscuit_amd.weights import synthetic_weights
w = synthetic_weights(1)
for ns in (1, 2, 3):
    engs = [Engine(w, dtype='bf16', max_batch=256, max_mc=30) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    tiles = [torch.randint(0, 256, (256, 299, 299, 3), dtype=torch.uint8, device='cuda') for _ in range(ns)]
    outs = [(torch.empty((256, 2), device='cuda'), torch.empty((256, 2), device='cuda')) for _ in range(ns)]
    def run(steps):
        for i in range(steps):
            k = i % ns
            with torch.cuda.stream(streams[k]):
                engs[k].mc_infer(tiles[k], 30, 1234, tile_idx0=i * 256, out=outs[k])
    run(2 * ns); torch.cuda.synchronize()
    t = time.perf_counter(); steps = 24
    run(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'streams={ns}: {dt/steps*1e3:.2f} ms/batch  {steps*256/dt:.0f} tiles/s', flush=True)
    del engs
