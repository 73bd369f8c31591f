"""Generate golden vectors for the CONSUMER of the hot path by importing the
reference's own ``biscuit/threshold.py`` / ``biscuit/utils.py`` in the build container.

TEST INFRASTRUCTURE ONLY.  Runs only where ``/root/reference`` exists (never on the GPU
box); its output ``tests/golden/consumer_*.json.gz`` is committed data: inputs (seeded
synthetic tile-prediction frames) and the reference's outputs for
``process_tile_predictions`` (threshold.py:125-177), ``process_group_predictions``
(threshold.py:180-245), ``apply`` (248-361), ``detect`` (364-475), ``from_cv`` (478-557).

The reference imports slideflow / seaborn / skmisc at module level; none is installed,
so three stub modules providing only the names touched at import time
(``slideflow.util.log``, ``seaborn``, ``skmisc.loess``) are registered first and
``biscuit/__init__.py`` is not executed.  No reference source is copied.

usage: python oracle/make_consumer_golden.py [--ref /root/reference]
"""
import argparse
import gzip
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np
import pandas as pd
import sklearn


def import_reference(ref):
    sf = types.ModuleType('slideflow')
    sf_util = types.ModuleType('slideflow.util')
    sf_util.log = logging.getLogger('slideflow-stub')
    sf_util.path_to_ext = lambda p: os.path.splitext(p)[1][1:]
    sf.util = sf_util
    sys.modules['slideflow'] = sf
    sys.modules['slideflow.util'] = sf_util
    sys.modules['seaborn'] = types.ModuleType('seaborn')
    sk = types.ModuleType('skmisc')
    sk.loess = types.ModuleType('skmisc.loess')
    sk.loess.loess = None   # only used by plot_uncertainty (threshold.py:101), never called here
    sys.modules['skmisc'] = sk
    sys.modules['skmisc.loess'] = sk.loess
    pkg = types.ModuleType('biscuit')
    pkg.__path__ = [os.path.join(ref, 'biscuit')]
    sys.modules['biscuit'] = pkg
    mods = {}
    for name in ('errors', 'delong', 'utils', 'threshold'):
        spec = importlib.util.spec_from_file_location(
            f'biscuit.{name}', os.path.join(ref, 'biscuit', f'{name}.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'biscuit.{name}'] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    return mods


def make_frame(seed, n_slides, tiles_per_slide, flip_frac=0.2, ragged=False):
    """Seeded synthetic tile-prediction frame: y_pred correlates with the slide label,
    a fraction of slides is mislabelled (so slide-level UQ thresholds are detectable) and
    uncertainty is larger for wrong / ambiguous tiles."""
    rng = np.random.default_rng(seed)
    rows = []
    for s in range(n_slides):
        y = s % 2
        wrong = rng.random() < flip_frac
        centre = (0.72 if (y ^ wrong) else 0.28) + rng.normal(0, 0.06)
        nt = tiles_per_slide if not ragged else int(rng.integers(3, tiles_per_slide + 1))
        yp = np.clip(centre + rng.normal(0, 0.17, nt), 0.001, 0.999)
        amb = 1.0 - 2.0 * np.abs(yp - 0.5)
        unc = np.clip(0.004 + 0.035 * amb * rng.uniform(0.3, 1.0, nt) + (0.012 if wrong else 0.0)
                      * rng.uniform(0.5, 1.5, nt), 1e-4, 0.5)
        for t in range(nt):
            rows.append((f'slide{s:03d}', f'patient{s // 2:03d}' if s % 5 == 0 else f'patient_s{s:03d}',
                         y, float(yp[t]), float(unc[t])))
    # interleave slides so first-appearance order != sorted order
    df = pd.DataFrame(rows, columns=['slide', 'patient', 'y_true', 'y_pred', 'uncertainty'])
    order = rng.permutation(len(df))
    return df.iloc[order].reset_index(drop=True)


def jsonable(x):
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (np.floating, float)):
        return None if np.isnan(x) else float(x)
    if isinstance(x, (np.integer, int, np.bool_, bool)):
        return int(x)
    if isinstance(x, np.ndarray):
        return jsonable(x.tolist())
    return x


def frame_out(df, cols):
    return {c: jsonable(df[c].to_numpy()) for c in cols if c in df.columns}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    args = ap.parse_args()
    m = import_reference(args.ref)
    th = m['threshold']
    os.makedirs(args.out, exist_ok=True)
    cases = []
    specs = [dict(seed=11, n_slides=16, tiles_per_slide=64, flip_frac=0.0),   # config-1 shape
             dict(seed=12, n_slides=40, tiles_per_slide=64, flip_frac=0.25),
             dict(seed=13, n_slides=24, tiles_per_slide=40, flip_frac=0.3, ragged=True)]
    gcols = ['error', 'uncertainty', 'correct', 'incorrect', 'y_true', 'y_pred', 'y_pred_bin']
    for spec in specs:
        df = make_frame(**spec)
        case = {'spec': spec,
                'input': {c: jsonable(df[c].to_numpy()) for c in df.columns}}
        patients = dict(zip(df['slide'], df['patient']))
        for pt in (0.5, 'detect'):
            d, t = th.process_tile_predictions(df.copy(), pred_thresh=pt, patients=patients)
            case[f'tile_{pt}'] = {'pred_thresh': jsonable(t),
                                  'cols': frame_out(d, ['error', 'correct', 'incorrect', 'y_pred_bin', 'patient'])}
        d, _ = th.process_tile_predictions(df.copy(), pred_thresh=0.5, patients=patients)
        for level in ('slide', 'patient'):
            for pt in (0.5, 'detect'):
                g, t = th.process_group_predictions(d.copy(), pred_thresh=pt, level=level)
                case[f'group_{level}_{pt}'] = {'pred_thresh': jsonable(t), 'levels': list(g[level]),
                                               'cols': frame_out(g, gcols)}
        tile_uq = float(np.quantile(df['uncertainty'], 0.7))
        g, _ = th.process_group_predictions(d[d['uncertainty'] < tile_uq].copy(), pred_thresh=0.5, level='slide')
        case['group_slide_filtered'] = {'tile_uq': tile_uq, 'levels': list(g['slide']),
                                        'cols': frame_out(g, gcols)}
        thr, auc = th.detect(df.copy(), patients=patients)
        case['detect'] = {'thresholds': jsonable(thr), 'auc': jsonable(auc)}
        for level in ('slide', 'patient'):
            slide_uq = float(np.quantile(g['uncertainty'], 0.8))
            res, s_df = th.apply(df.copy(), tile_uq=tile_uq, slide_uq=slide_uq, tile_pred=0.5,
                                 slide_pred=0.5, patients=patients, level=level)
            case[f'apply_{level}'] = {'tile_uq': tile_uq, 'slide_uq': slide_uq, 'results': jsonable(res),
                                      'levels': list(s_df[level]), 'cols': frame_out(s_df, gcols)}
        res, s_df = th.apply(df.copy(), tile_uq=0.0, slide_uq=0.0, patients=patients)  # 0 disables (threshold.py:297,323); None breaks its log f-string (threshold.py:284)
        case['apply_nofilter'] = {'results': jsonable(res), 'levels': list(s_df['slide'])}
        cases.append(case)
    folds = [make_frame(seed=20 + k, n_slides=40, tiles_per_slide=48, flip_frac=0.25) for k in range(3)]
    cv = {'folds': [{c: jsonable(f[c].to_numpy()) for c in f.columns} for f in folds]}
    cv['from_cv_detect'] = jsonable(th.from_cv([f.copy() for f in folds], tile_uq='detect', slide_uq='detect',
                                               tile_pred='detect', slide_pred='detect'))
    first = th.from_cv([f.copy() for f in folds], tile_uq='detect', slide_uq=None, tile_pred='detect',
                       slide_pred='detect')
    cv['from_cv_tile_only'] = jsonable(first)
    cv['from_cv_second'] = jsonable(th.from_cv([f.copy() for f in folds], tile_uq=float(first['tile_uq']),
                                               slide_uq='detect', tile_pred='detect', slide_pred='detect'))
    clean = make_frame(seed=31, n_slides=16, tiles_per_slide=32, flip_frac=0.0)
    clean['y_pred'] = np.where(clean['y_true'] == 1, 0.9, 0.1) + 0.0 * clean['y_pred']
    try:
        th.from_cv([clean.copy()], tile_uq='detect', slide_uq='detect', tile_pred='detect', slide_pred='detect')
        cv['from_cv_clean_raises'] = None
    except Exception as e:  # noqa: BLE001 -- record whatever the reference raises
        cv['from_cv_clean_raises'] = type(e).__name__
    cv['clean'] = {c: jsonable(clean[c].to_numpy()) for c in clean.columns}
    nan_df = folds[0].copy()
    nan_df.loc[3, 'y_pred'] = np.nan
    try:
        th.process_tile_predictions(nan_df)
        cv['nan_raises'] = None
    except Exception as e:  # noqa: BLE001
        cv['nan_raises'] = type(e).__name__
    meta = {'generator': 'oracle/make_consumer_golden.py', 'reference': 'jamesdolezal/biscuit @ 2024_10_08',
            'versions': {'numpy': np.__version__, 'pandas': pd.__version__, 'sklearn': sklearn.__version__}}
    with gzip.open(os.path.join(args.out, 'consumer_cases.json.gz'), 'wt') as f:
        json.dump({'meta': meta, 'cases': cases}, f)
    with gzip.open(os.path.join(args.out, 'consumer_cv.json.gz'), 'wt') as f:
        json.dump({'meta': meta, 'cv': cv}, f)
    print('wrote', args.out, [os.path.getsize(os.path.join(args.out, n))
                              for n in ('consumer_cases.json.gz', 'consumer_cv.json.gz')])


if __name__ == '__main__':
    main()
