"""BASELINE.json config 4: MC-dropout sweep N in {1,5,10,30,50}: fused on-device Welford (backbone once,
N head passes in one launch sequence) vs N separate full passes, 1 GPU, batch 256, f16 (the headline storage type).
Prints one JSON line per (N, mode).  usage: python tools/sweep_mc.py [--batch 256]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd.engine import Engine
from biscuit_amd.weights import synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--steps', type=int, default=8)
args = ap.parse_args()
ap_dtype = 'f16'
eng = Engine(synthetic_weights(1), dtype=ap_dtype, max_batch=args.batch, max_mc=50)
tiles = torch.randint(0, 256, (args.batch, 299, 299, 3), dtype=torch.uint8, device='cuda')
for mc in (1, 5, 10, 30, 50):
    ref = None
    for mode in ('head', 'full'):
        steps = args.steps if mode == 'head' else max(1, args.steps // 4)
        out = eng.mc_infer(tiles, mc, 1234, mc_mode=mode)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            out = eng.mc_infer(tiles, mc, 1234, mc_mode=mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        same = None
        if ref is None:
            ref = out
        else:
            same = bool(torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1]))
        print(json.dumps({'mc_n': mc, 'mode': mode, 'ms_per_batch': dt * 1e3, 'tiles_per_s': args.batch / dt,
                          'bit_identical_to_head': same}), flush=True)
