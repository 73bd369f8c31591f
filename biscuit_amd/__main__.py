"""Command-line entry to the hot path: MC-dropout inference over slides, tile table on disk,
slide-level table and (optionally) thresholded metrics -- what ``Project.evaluate(...,
save_predictions=True)`` followed by ``threshold.apply`` does in the reference
(``biscuit/experiment.py:917-922``, ``705-720``).

    python -m biscuit_amd --tfrecords DIR --labels labels.csv --weights model.npz --out eval_dir
    python -m biscuit_amd --synthetic 16x64 --out eval_dir          # random-init weights, synthetic tiles

labels.csv: columns ``slide,label[,patient]`` (label 0/1).  Needs an MI355X; there is no CPU path.
"""
import argparse
import glob
import json
import os

import numpy as np
import pandas as pd


def model_hp(params):
    """(ModelParams, norm_fit) from what ``keras_import.read_params`` found in a model's params.json: the dropout
    rate (and ``uq_n`` when stored) drive the MC statistics, so they come from the model, never from the defaults;
    a stain normaliser other than the one this path implements (`reinhard_fast`, hp.py:19) is an error, and its fit
    is used only when that normaliser is the model's."""
    from .hp import ModelParams
    hp = ModelParams()
    if not params:
        return hp, None
    raw = params.get('hp') or {}
    if raw.get('dropout') is not None:
        hp.dropout = float(raw['dropout'])
    if raw.get('uq_n') is not None:
        hp.uq_n = int(raw['uq_n'])
    normalizer = params.get('normalizer')
    if normalizer is None and params.get('norm_fit'):
        raise SystemExit(f"{params.get('path', 'params.json')}: a norm_fit block but no hp.normalizer: cannot tell which stain "
                         f"normaliser the model was trained with")
    hp.normalizer = normalizer
    hp.validate()
    if normalizer not in (None, 'reinhard_fast'):
        raise SystemExit(f"{params.get('path', 'params.json')}: the model was trained with normalizer={normalizer!r}; this "
                         f"path implements 'reinhard_fast' (biscuit/hp.py:19) or none")
    fit = params.get('norm_fit') if normalizer == 'reinhard_fast' else None
    if normalizer == 'reinhard_fast' and not fit:
        raise SystemExit(f"{params.get('path', 'params.json')}: normalizer='reinhard_fast' but no norm_fit block")
    return hp, fit


def _first_tiles(slide, n):
    """The first ``n`` decoded tiles of a slide as a uint8 array, reading no more than those: through the slide's chunk source
    when it has one (TFRecords: ``n`` records, not the slide -- a 10^4-tile slide is 2.7 GB decoded), with the PNG filters
    reversed on the host whatever mode the run itself uses."""
    if slide is None:
        return None
    cnt = min(int(n), slide.n_tiles)
    src = getattr(slide, 'source', None)
    if src is not None and hasattr(src, 'path'):
        from .inference import TFRecordSource
        one = TFRecordSource(src.path, slide.n_tiles, src.tile_px, rows=False)
        buf = np.empty(one.chunk_shape(cnt), np.uint8)
        try:
            one.read(0, cnt, buf)
        finally:
            one.close()
        return buf
    t = slide.load()
    if hasattr(t, 'rows'):          # filtered PNG scanlines without a chunk source: not tiles yet
        return None
    t = t[:cnt]
    return np.ascontiguousarray(t.cpu().numpy() if hasattr(t, 'cpu') else t)


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m biscuit_amd', description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument('--tfrecords', help='directory of Slideflow *.tfrecords (one per slide)')
    src.add_argument('--synthetic', help='SxT: S synthetic slides of T tiles; or a comma list of tile counts (9,3,0,7)')
    ap.add_argument('--labels', help='CSV with slide,label[,patient]')
    ap.add_argument('--weights', help='weights: .npz / .safetensors under Keras variable names, a TensorFlow checkpoint '
                                      'prefix, or a Keras SavedModel directory (default: seeded random init)')
    ap.add_argument('--model', help='Slideflow model directory (find_model, biscuit/utils.py:233-272): SavedModel weights '
                                    'plus params.json (norm_fit, outcome labels); replaces --weights and --params')
    ap.add_argument('--outcome', default='cohort')
    ap.add_argument('--out', required=True)
    ap.add_argument('--mc', type=int, default=None, help='MC-dropout passes (default: uq_n of the model, 30)')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'])
    ap.add_argument('--streams', type=int, default=1,
                    help='batches in flight, each on its share of the chip (1 measured fastest through evaluate(): 25.8 k vs 25.4 k tiles/s)')
    ap.add_argument('--params', help="Slideflow params.json: its norm_fit switches on the reinhard_fast stain normaliser (hp.py:19)")
    ap.add_argument('--tile-uq', type=float, default=0.0, help='tile-level uncertainty threshold (0 = off)')
    ap.add_argument('--slide-uq', type=float, default=0.0, help='slide-level uncertainty threshold (0 = off)')
    ap.add_argument('--gpu-decode', type=int, default=0, metavar='CUS',
                    help='decode PNG tiles on the GPU: the host only copies their zlib streams, CUS compute units (16-32) kept out of '
                         'the inference streams inflate them.  For hosts with few free cores per GPU and runs of >= 30 k tiles; '
                         'slower than 16 host threads otherwise (profiles/r05_inflate.txt).  Pins 3 x up to 1 GiB of host memory per rank for the '
                         'compressed ring (4 096-tile chunks) and allocates ~3 GB on the device per chunk in flight')
    ap.add_argument('--skip-existing', action='store_true',
                    help='do nothing when OUT already holds tile_predictions_eval.csv (the idempotence of the reference\'s Step 6: '
                         'utils.eval_exists, biscuit/experiment.py:913-914)')
    ap.add_argument('--detect', action='store_true',
                    help='also run threshold.detect on the tile table (Youden thresholds over every tile of the cohort, threshold.py:364-475)')
    ap.add_argument('--dist-backend', default=None, help='process-group backend for WORLD_SIZE > 1 (default nccl = RCCL; gloo for a rehearsal '
                                                          'of several ranks on one GPU)')
    ap.add_argument('--local-device', type=int, default=None, help='HIP device of this rank (default LOCAL_RANK)')
    ap.add_argument('--unfilter', default='auto', choices=['auto', 'host', 'gpu'],
                    help="where the PNG scanline filters of TFRecord tiles are reversed: 'auto' (default) decodes 48 tiles of the first "
                         "slide both ways and takes the GPU where the host alone cannot feed it (same bytes either way; 28.8 k -> 31.9 k "
                         "tiles/s on 16 cores)")
    ap.add_argument('--no-calibrate', action='store_true',
                    help='f16 with external weights: skip the activation-exponent calibration on the first tiles (the headroom check stays)')
    args = ap.parse_args(argv)

    if args.skip_existing and os.path.exists(os.path.join(args.out, 'tile_predictions_eval.csv')):
        # (every rank sees the same directory and takes the same way out, before any rendezvous)
        if int(os.environ.get('RANK', '0')) == 0:
            print(json.dumps({'skipped': True, 'tile_table': os.path.join(args.out, 'tile_predictions_eval.csv')}))
        return
    from . import distributed as D, threshold, weights as W
    from .engine import EnginePool
    from .inference import Slide, evaluate, slides_from_tfrecords
    from .synthetic import make_slides

    rank, world, local = D.init_from_env('cuda', backend=args.dist_backend, local_device=args.local_device)
    D.pin_rank(int(os.environ.get('LOCAL_RANK', rank)), int(os.environ.get('LOCAL_WORLD_SIZE', world)), device_index=local)     # before any thread pool exists
    model_params = None
    if args.model:
        from .keras_import import load_model_dir
        w, model_params = load_model_dir(args.model)
    elif args.weights:
        from .keras_import import load_weights
        w = load_weights(args.weights)
    else:
        w = W.synthetic_weights(1)
    patients = None
    if args.tfrecords:
        lab = pd.read_csv(args.labels, dtype={'slide': str}) if args.labels else pd.DataFrame(columns=['slide', 'label'])
        labels = dict(zip(lab['slide'], lab['label']))
        patients = dict(zip(lab['slide'], lab['patient'])) if 'patient' in lab.columns else None
        paths = sorted(glob.glob(os.path.join(args.tfrecords, '*.tfrecords')))
        slides = slides_from_tfrecords(paths, labels, patients, gpu_decode=args.gpu_decode > 0,
                                       gpu_unfilter={'auto': 'auto', 'host': False, 'gpu': True}[args.unfilter])
    else:
        if 'x' in args.synthetic.lower():
            s, t = (int(x) for x in args.synthetic.lower().split('x'))
            tiles, sidx, y = make_slides(s, t, seed=0)
            slides = [Slide(f'slide{i:03d}', tiles[sidx == i], t, y_true=int(y[i])) for i in range(s)]
        else:                                                     # explicit (ragged) tile counts: 9,3,0,7
            from .synthetic import make_tiles
            counts = [int(x) for x in args.synthetic.split(',')]
            slides = [Slide(f'slide{i:03d}', make_tiles(c, seed=500 + i, slide_bias=[(i % 2) * 60.0 - 30.0, 0.0, (i % 3) * 10.0]), c,
                            y_true=i % 2) for i, c in enumerate(counts)]
    hp, norm_fit = model_hp(model_params)
    if args.params:
        with open(args.params) as f:
            raw = json.load(f)
        hp, norm_fit = model_hp({'hp': raw.get('hp') or {}, 'normalizer': (raw.get('hp') or {}).get('normalizer', raw.get('normalizer', 'reinhard_fast')),
                                 'norm_fit': raw.get('norm_fit'), 'path': args.params})
        if not norm_fit:
            raise SystemExit(f'{args.params}: no norm_fit block')
    if args.mc is None:
        args.mc = hp.uq_n
    act_exp, probe = None, None
    if args.dtype == 'f16' and (args.model or args.weights) and slides:
        # IEEE half ends at 65504 and every f16 kernel clamps there without a signal (MODE.FP16_OVFL).  With weights from outside
        # the storage is kept in range by construction: the first tiles go through the fp32 kernels once, every stored tensor's
        # peak is measured, and power-of-two activation exponents are folded into the BatchNorm constants (weights.py).  Every
        # rank reads the same 16 tiles of the same slide, so every rank packs the same blob.
        probe = _first_tiles(next((s for s in slides if s.n_tiles), None), 16)
        if probe is not None and not args.no_calibrate:
            from .engine import Engine
            act_exp, peaks = Engine.calibrate(w, probe, hp=hp, device=local, norm_fit=norm_fit)
            if rank == 0 and any(act_exp.values()):
                big = {t: k for t, k in act_exp.items() if k}
                print(f'f16: activation exponents from the first {len(probe)} tiles (peak {max(peaks.values()):.3g}): {big}', flush=True)
    pool = EnginePool(w, n_streams=args.streams, hp=hp, dtype=args.dtype, max_batch=args.batch, max_mc=args.mc, device=local,
                      act_exp=act_exp, reserve_cus=max(0, args.gpu_decode))
    if probe is not None:
        import torch
        t = torch.from_numpy(probe[:8]).to(pool.engines[0].device)
        if norm_fit is not None:
            t = pool.engines[0].reinhard_fast(t, norm_fit['target_means'], norm_fit['target_stds'])
        hr = pool.engines[0].f16_headroom(t)                      # the check behind the construction
        if any(hr['saturated'].values()):
            raise SystemExit(f'f16 storage saturates with these weights ({hr["saturated"]}): run with --dtype bf16 or f32')
        if hr['headroom'] < 8 and rank == 0:
            print(f'warning: f16 headroom only {hr["headroom"]:.1f}x on the first tiles ({hr["max_abs"]}); '
                  f'consider --dtype bf16', flush=True)
    # the tile table streams to disk while the GPU works (every rank its shard; rank 0 splices them after the gather): the frame
    # is not kept in memory, the consumer reads the file -- as biscuit does (experiment.py:688-699)
    res = evaluate(pool, slides, outcome=args.outcome, mc_n=args.mc, seed=args.seed, batch=args.batch,
                   save_dir=args.out, rank=rank, world=world, norm_fit=norm_fit, keep_tiles=False)
    if rank == 0:
        from .predictions import load_tile_predictions
        sf, _ = res.slide_frame(0.5)
        sf.to_csv(os.path.join(args.out, f'slide_predictions_{args.outcome}_eval.csv'), index=False)
        summary = {'slides': int((res.slide_count > 0).sum()), 'tiles': int(res.slide_count.sum()), 'world': world,
                   'tile_table': res.table_path}
        # the consumer on THE table, at any world size: rank 0 holds every tile of the cohort once the shards are spliced
        # (threshold.detect takes Youden's J over all of them, threshold.py:417-426)
        df = load_tile_predictions(res.table_path, args.outcome)
        if args.detect:
            try:
                found, auc = threshold.detect(df.copy(), patients=patients)
                summary['detected'] = {k: (None if v is None else float(v)) for k, v in found.items()}
                summary['detect_auc'] = None if auc is None or np.isnan(auc) else float(auc)
            except ValueError as e:                               # (a cohort whose tiles are all correct / of one class)
                summary['detected'], summary['detect_error'] = None, str(e)
        metrics, _ = threshold.apply(df, tile_uq=args.tile_uq, slide_uq=args.slide_uq, patients=patients)
        summary.update({k: (None if v is None or (isinstance(v, float) and np.isnan(v)) else float(v))
                        for k, v in metrics.items()})
        print(json.dumps(summary))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
