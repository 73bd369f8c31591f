"""Developer probe run on the GPU box through gpurun: per-stage parity vs the CPU oracle
and quick timings.  Not part of the product or the test-suite.

usage: python tools/gpu_probe.py {stage|head|layers|e2e|time} [--dtype f32|bf16] [--n N]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from biscuit_amd.engine import Engine  # noqa: E402
from biscuit_amd.synthetic import make_slides  # noqa: E402
from biscuit_amd.weights import synthetic_weights  # noqa: E402
from oracle.xception_ref import XceptionOracle, standardize  # noqa: E402

TAPS = [('staged', (299, 299, 3)), ('block1_conv1', (149, 149, 32)), ('block1_conv2', (147, 147, 64)),
        ('block2_res', (74, 74, 128)), ('block2_sepconv1', (147, 147, 128)),
        ('block2_sepconv2', (147, 147, 128)), ('block2_out', (74, 74, 128)),
        ('block3_res', (37, 37, 256)), ('block3_sepconv1', (74, 74, 256)),
        ('block3_sepconv2', (74, 74, 256)), ('block3_out', (37, 37, 256)),
        ('block4_res', (19, 19, 728)), ('block4_sepconv1', (37, 37, 728)),
        ('block4_sepconv2', (37, 37, 728)), ('block4_out', (19, 19, 728))] + \
       [(f'block{b}_out', (19, 19, 728)) for b in range(5, 13)] + \
       [('block13_out', (10, 10, 1024)), ('block14_sepconv1', (10, 10, 1536)),
        ('block14_sepconv2', (10, 10, 2048))]


def err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    return f'max|d|={d.max():.3e} rms={np.sqrt((d**2).mean()):.3e} ref_rms={np.sqrt((b**2).mean()):.3e} nan={int(np.isnan(a).sum())}'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--n', type=int, default=2)
    ap.add_argument('--mc', type=int, default=5)
    args = ap.parse_args()
    torch.manual_seed(0)
    w = synthetic_weights(1)
    tiles, sidx, _ = make_slides(max(1, args.n // 2), 2 if args.n > 1 else 1, seed=0)
    tiles = tiles[:args.n]
    dev = torch.device('cuda:0')
    eng = Engine(w, dtype=args.dtype, max_batch=max(args.n, 256), max_mc=50)
    orc = XceptionOracle(w, emulate_bf16=(args.dtype == 'bf16'))
    d_tiles = torch.from_numpy(tiles).to(dev)

    if args.what == 'stage':
        out = eng.stage(d_tiles).float().cpu().numpy()
        ref = standardize(tiles).numpy()
        if args.dtype == 'bf16':
            ref = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
        print('stage', err(out, ref))
    elif args.what == 'head':
        feat = np.load('/dev/null') if False else np.abs(np.random.default_rng(3).normal(0.8, 0.5, (args.n, 2048))).astype(np.float32)
        m, s = eng.mc_head(torch.from_numpy(feat).to(dev), args.mc, 1234, tile_idx0=7)
        rm, rs = orc.mc_from_features(feat, args.mc, 1234, tile_index0=7)
        print('head mean', err(m.cpu().numpy(), rm))
        print('head std ', err(s.cpu().numpy(), rs))
        print(m.cpu().numpy()[:3], rm[:3]); print(s.cpu().numpy()[:3], rs[:3])
    elif args.what == 'layers':
        staged_ref = standardize(tiles)
        taps = {}
        feat_ref = orc.backbone(staged_ref, taps)
        staged = eng.stage(d_tiles)
        for name, shp in TAPS:
            got = eng.debug_activation(name, staged, shp).cpu().numpy()
            ref = taps[name].permute(0, 2, 3, 1).numpy()
            print(f'{name:18s}', err(got, ref), flush=True)
        feat = eng.backbone(staged).cpu().numpy()
        print('features          ', err(feat, feat_ref.numpy()))
    elif args.what == 'e2e':
        m, s = eng.mc_infer(d_tiles, args.mc, 1234, tile_idx0=0, mc_mode='head')
        m2, s2 = eng.mc_infer(d_tiles, args.mc, 1234, tile_idx0=0, mc_mode='full')
        rm, rs = orc.mc_predict(tiles, args.mc, 1234, mode='head')
        print('e2e mean', err(m.cpu().numpy(), rm))
        print('e2e std ', err(s.cpu().numpy(), rs))
        print('head==full', bool((m == m2).all().item()), bool((s == s2).all().item()))
    elif args.what == 'time':
        n = args.n
        big = torch.randint(0, 256, (n, 299, 299, 3), dtype=torch.uint8, device=dev)
        for _ in range(2):
            eng.mc_infer(big, 30, 1234)
        torch.cuda.synchronize()
        t = time.time()
        reps = 5
        for _ in range(reps):
            eng.mc_infer(big, 30, 1234)
        torch.cuda.synchronize()
        dt = (time.time() - t) / reps
        print(f'time dtype={args.dtype} n={n}: {dt*1e3:.2f} ms/batch  {n/dt:.1f} tiles/s')
        eng.profile_enable(True)
        for _ in range(2):
            eng.mc_infer(big, 30, 1234)
        ents = eng.profile_read()
        eng.profile_enable(False)
        tot = sum(e.ms for e in ents)
        for e in sorted(ents, key=lambda e: -e.ms):
            per = e.ms / max(e.launches, 1)
            print(f'{e.name:36s} n={e.launches:4d} ms/launch={per:8.4f} share={e.ms/tot*100:5.1f}% '
                  f'TF/s={e.flops/per/1e9:9.1f} GB/s={e.bytes/per/1e6:9.1f}')
    else:
        raise SystemExit('unknown command')


if __name__ == '__main__':
    main()
