"""Tile reader, with and without the GPU un-filter (GPU box): host decode rate of PNG TFRecords as decoded tiles and as
filtered rows, the kernel's time for a batch of 256, and `evaluate` end to end both ways -- for noise-like tiles (the
bench's synthetic ones: the encoder picks filter None) and photo-like ones (smooth + grain: Sub / Up / Average / Paeth rows)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from biscuit_amd import tfrecord, tfrecord_native as tn      # noqa: E402
from biscuit_amd.engine import Engine                        # noqa: E402
from biscuit_amd.inference import evaluate, slides_from_tfrecords   # noqa: E402
from biscuit_amd.synthetic import make_tiles                 # noqa: E402
from biscuit_amd.weights import synthetic_weights            # noqa: E402


def photo(px, seed):
    r = np.random.default_rng(seed)
    y, x = np.mgrid[0:px, 0:px]
    base = np.stack([150 + 60 * np.sin(x / (11.0 + seed % 7) + c) + 30 * np.cos(y / (7.0 + seed % 5) * (c + 1) / 2) for c in range(3)], -1)
    blobs = 40 * np.exp(-(((x - 150 + 9 * (seed % 11)) ** 2 + (y - 140) ** 2) / 3000.0))[..., None]
    return np.clip(base - blobs + r.normal(0, 5, base.shape), 0, 255).astype(np.uint8)


def main():
    d = tempfile.mkdtemp(prefix='bq_rd_')
    n_slides, per = 6, 256
    eng = Engine(synthetic_weights(1), dtype='f16', max_batch=256, max_mc=30)
    for kind in ('noise', 'photo'):
        base = make_tiles(32, seed=21) if kind == 'noise' else np.stack([photo(299, s) for s in range(32)])
        enc = [tfrecord.encode_image(t) for t in base]
        paths = []
        for s in range(n_slides):
            p = os.path.join(d, f'{kind}{s}.tfrecords')
            tfrecord.write_slide(p, f'{kind}{s}', [enc[(i + s) % 32] for i in range(per)], np.zeros((per, 2), np.int64))
            paths.append(p)
        with tn.NativeReader(paths[0]) as r:
            rows, _ = r.decode(rows=True)
        hist = np.bincount(rows[:, :, 0].ravel(), minlength=5) / rows[:, :, 0].size
        res = {'png_bytes_per_tile': os.path.getsize(paths[0]) / per, 'filter_mix_none_sub_up_avg_paeth': [round(float(h), 3) for h in hist]}
        for mode in (False, True):
            for p in paths[:1]:
                tfrecord.read_slide(p, 299, rows=mode)
            t0 = time.perf_counter()
            for p in paths:
                tfrecord.read_slide(p, 299, rows=mode)
            res['decode_rows_tiles_per_s' if mode else 'decode_full_tiles_per_s'] = n_slides * per / (time.perf_counter() - t0)
        dr = torch.from_numpy(rows).cuda()
        eng.png_unfilter(dr); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.png_unfilter(dr)
        e1.record(); torch.cuda.synchronize()
        res['unfilter_ms_per_256_tiles'] = e0.elapsed_time(e1) / 10
        dr4 = dr.repeat(4, 1, 1).contiguous()                      # a 1 024-tile slide: four waves per CU side by side
        eng.png_unfilter(dr4); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            eng.png_unfilter(dr4)
        e1.record(); torch.cuda.synchronize()
        res['unfilter_ms_per_1024_tiles'] = e0.elapsed_time(e1) / 5
        labels = {f'{kind}{s}': s % 2 for s in range(n_slides)}
        for rnd in range(3):                                       # alternate the two modes: the first pass of either pays for cold caches
            for mode in (False, True):
                sl = slides_from_tfrecords(paths, labels, gpu_unfilter=mode)
                t0 = time.perf_counter()
                evaluate(eng, sl, mc_n=30, seed=1, batch=256, keep_tiles=False)
                res[f"evaluate_{'gpu' if mode else 'host'}_unfilter_tiles_per_s_round{rnd}"] = n_slides * per / (time.perf_counter() - t0)
        print(kind, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}, flush=True)
    print('host cores', len(os.sched_getaffinity(0)))


if __name__ == '__main__':
    main()
