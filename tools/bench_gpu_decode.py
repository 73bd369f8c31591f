"""TFRecords -> slide results end to end, host decode vs device decode (``slides_from_tfrecords(gpu_decode=True)``), at a run length
where the device path reaches its steady state (a compressed chunk is 4 096 tiles; a stream is inflated by ONE lane, 100-300 ms a
chunk whatever its size, so the path needs many chunks in flight and a run of many chunks to show its rate).

    python tools/bench_gpu_decode.py [--cores 32] [--slides 64] [--reserve 16 24 32] [--kind noise photo]

--cores N pins the process to N host cores first: 32 = this rank's share of a 256-core, 8-GPU node.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cores', type=int, default=0)
    ap.add_argument('--slides', type=int, default=64)
    ap.add_argument('--per', type=int, default=1024)
    ap.add_argument('--reserve', type=int, nargs='*', default=[16, 24, 32])
    ap.add_argument('--kind', nargs='*', default=['noise', 'photo'])
    ap.add_argument('--mc', type=int, default=30)
    ap.add_argument('--decode-streams', type=int, default=2)
    args = ap.parse_args()
    if args.cores:
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.cores])
    import numpy as np
    from biscuit_amd import tfrecord, tfrecord_native
    from biscuit_amd.engine import EnginePool
    from biscuit_amd.inference import evaluate, slides_from_tfrecords
    from biscuit_amd.synthetic import make_tiles
    from biscuit_amd.weights import synthetic_weights
    tfrecord_native.load()
    print(f'host cores {len(os.sched_getaffinity(0))}, decoder threads {tfrecord_native.default_threads()}', flush=True)
    w = synthetic_weights(1)
    n = args.slides * args.per
    d = tempfile.mkdtemp(prefix='bq_gd_')
    try:
        for kind in args.kind:
            base = [tfrecord.encode_image(t) for t in make_tiles(32, seed=21, grain=18.0 if kind == 'noise' else 4.0)]
            paths = []
            for s in range(args.slides):
                p = os.path.join(d, f'{kind}{s}.tfrecords')
                if s < 8:
                    tfrecord.write_slide(p, f'{kind}{s}', [base[(i + s) % 32] for i in range(args.per)], np.zeros((args.per, 2), np.int64))
                else:
                    os.symlink(os.path.join(d, f'{kind}{s % 8}.tfrecords'), p)      # (the same bytes again: the page cache holds 8 files)
                paths.append(p)
            lab = {f'{kind}{s}': s % 2 for s in range(args.slides)}

            def rate(pool, slides, batch):
                evaluate(pool, slides[:8], mc_n=args.mc, seed=1, batch=batch, keep_tiles=False)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    r = evaluate(pool, slides, mc_n=args.mc, seed=1, batch=batch, keep_tiles=False)
                    ts.append(time.perf_counter() - t0)
                return n / sorted(ts)[1], r

            pool = EnginePool(w, n_streams=1, dtype='f16', max_batch=256, max_mc=args.mc)
            host, ref = rate(pool, slides_from_tfrecords(paths, lab), 256)
            rows, _ = rate(pool, slides_from_tfrecords(paths, lab, gpu_unfilter=True), 256)
            pool.synchronize(); pool.close()
            print(f'{kind}: {n} tiles; host decode {host:8.0f} tiles/s; host inflate + device un-filter {rows:8.0f}', flush=True)
            for rc in args.reserve:
                pool = EnginePool(w, n_streams=1, reserve_cus=rc, decode_streams=args.decode_streams, dtype='f16', max_batch=256, max_mc=args.mc)
                batch = (256 - rc) // 16 * 16
                z, r = rate(pool, slides_from_tfrecords(paths, lab, gpu_decode=True), batch)
                same = bool(np.array_equal(r.slide_pred, ref.slide_pred) and np.array_equal(r.slide_unc, ref.slide_unc))
                pool.synchronize(); pool.close()
                print(f'    device decode on {rc:3d} CUs ({args.decode_streams} streams), batch {batch}: {z:8.0f} tiles/s; results equal: {same}', flush=True)
            for p in paths:
                os.unlink(p)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
