"""biscuit_amd.threshold vs golden vectors captured from the reference's own
biscuit/threshold.py (oracle/make_consumer_golden.py).  Exact for integer/bool columns
and thresholds picked from data; 1e-12 for float64 means."""
import numpy as np
import pandas as pd
import pytest

from biscuit_amd import errors, threshold as th


def frame(d):
    return pd.DataFrame({k: (np.array(v, dtype=float) if k in ('y_pred', 'uncertainty') else v)
                         for k, v in d.items()})


def close(a, b, tol=1e-12):
    a = np.array([np.nan if x is None else x for x in np.ravel(a)], dtype=float)
    b = np.array([np.nan if x is None else x for x in np.ravel(b)], dtype=float)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=0, atol=tol, equal_nan=True)


def check_group(got, want, level):
    assert list(got[level]) == want['levels']            # first-appearance order
    for c in ('correct', 'incorrect', 'y_true', 'y_pred_bin'):
        assert [int(x) for x in got[c]] == want['cols'][c], c
    for c in ('error', 'uncertainty', 'y_pred'):
        close(got[c].to_numpy(), want['cols'][c])


def test_versions_recorded(consumer_cases):
    assert 'sklearn' in consumer_cases['meta']['versions']


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_process_tile_predictions(consumer_cases, ci):
    case = consumer_cases['cases'][ci]
    df0 = frame(case['input'])
    patients = dict(zip(df0['slide'], df0['patient']))
    for pt in (0.5, 'detect'):
        df = df0.drop(columns=['patient']).copy()
        out, t = th.process_tile_predictions(df, pred_thresh=pt, patients=patients)
        want = case[f'tile_{pt}']
        assert out is df                                     # mutates in place like the reference
        assert float(t) == want['pred_thresh']
        for c in ('correct', 'incorrect', 'y_pred_bin'):
            assert [int(x) for x in out[c]] == want['cols'][c]
        close(out['error'].to_numpy(), want['cols']['error'])
        assert list(out['patient']) == want['cols']['patient']


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_process_group_predictions(consumer_cases, ci):
    case = consumer_cases['cases'][ci]
    df, _ = th.process_tile_predictions(frame(case['input']), pred_thresh=0.5)
    for level in ('slide', 'patient'):
        for pt in (0.5, 'detect'):
            got, t = th.process_group_predictions(df.copy(), pred_thresh=pt, level=level)
            want = case[f'group_{level}_{pt}']
            assert float(t) == want['pred_thresh']
            check_group(got, want, level)
    want = case['group_slide_filtered']
    got, _ = th.process_group_predictions(df[df['uncertainty'] < want['tile_uq']].copy(), 0.5, 'slide')
    check_group(got, want, 'slide')


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_detect_and_apply(consumer_cases, ci):
    case = consumer_cases['cases'][ci]
    df0 = frame(case['input'])
    patients = dict(zip(df0['slide'], df0['patient']))
    thr, a = th.detect(df0.copy(), patients=patients)
    for k, v in case['detect']['thresholds'].items():
        assert (thr[k] is None and v is None) or float(thr[k]) == v, k
    close([a], [case['detect']['auc']])
    for level in ('slide', 'patient'):
        want = case[f'apply_{level}']
        res, s_df = th.apply(df0.copy(), tile_uq=want['tile_uq'], slide_uq=want['slide_uq'], tile_pred=0.5,
                             slide_pred=0.5, patients=patients, level=level)
        for k, v in want['results'].items():
            close([res[k]], [v])
        check_group(s_df.reset_index(drop=True), want, level)
    res, s_df = th.apply(df0.copy(), tile_uq=0.0, slide_uq=0.0, patients=patients)
    for k, v in case['apply_nofilter']['results'].items():
        close([res[k]], [v])
    assert list(s_df['slide']) == case['apply_nofilter']['levels']


def test_from_cv(consumer_cv):
    cv = consumer_cv['cv']
    folds = [frame(f) for f in cv['folds']]
    kw = dict(tile_uq='detect', slide_uq='detect', tile_pred='detect', slide_pred='detect')
    got = th.from_cv([f.copy() for f in folds], **kw)
    for k, v in cv['from_cv_detect'].items():
        assert float(got[k]) == pytest.approx(v, abs=1e-15), k
    first = th.from_cv([f.copy() for f in folds], tile_uq='detect', slide_uq=None, tile_pred='detect',
                       slide_pred='detect')
    for k, v in cv['from_cv_tile_only'].items():
        assert float(first[k]) == pytest.approx(v, abs=1e-15), k
    assert first['slide_uq'] == 0.5          # reference quirk: never None (threshold.py:461-463)
    second = th.from_cv([f.copy() for f in folds], tile_uq=float(first['tile_uq']), slide_uq='detect',
                        tile_pred='detect', slide_pred='detect')
    for k, v in cv['from_cv_second'].items():
        assert float(second[k]) == pytest.approx(v, abs=1e-15), k


def test_error_behaviour(consumer_cv):
    cv = consumer_cv['cv']
    # no incorrect tile at all: the reference's Youden search fails with ValueError
    assert cv['from_cv_clean_raises'] == 'ValueError'
    with pytest.raises(ValueError):
        th.from_cv([frame(cv['clean'])], tile_uq='detect', slide_uq='detect', tile_pred='detect',
                   slide_pred='detect')
    assert cv['nan_raises'] == 'PredsContainNaNError'
    bad = frame(cv['folds'][0])
    bad.loc[3, 'y_pred'] = np.nan
    with pytest.raises(errors.PredsContainNaNError):
        th.process_tile_predictions(bad)
    empty = frame(cv['folds'][0]).iloc[0:0]
    with pytest.raises(errors.ROCFailedError):
        th.process_group_predictions(empty, 0.5, 'slide')
    with pytest.raises(ValueError):
        th.process_group_predictions(frame(cv['folds'][0]).drop(columns=['uncertainty']), 0.5, 'slide')
    with pytest.raises(ValueError):
        th.from_cv([frame(cv['folds'][0]).drop(columns=['patient'])])


def test_rename_cols_against_reference_goldens():
    """The column contract: ``biscuit_amd.predictions.rename_cols`` leaves exactly the columns the
    reference's own ``utils.rename_cols`` (utils.py:31-53) left on the same frames -- dash and underscore spellings in
    every combination, the ``-y_true`` fallback, absent columns, explicit overrides, a non-string outcome
    (fixture: oracle/make_rename_golden.py)."""
    import json
    import os
    import pandas as pd
    from biscuit_amd.predictions import rename_cols
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rename_cols.json')))
    assert len(g['cases']) >= 90
    for c in g['cases']:
        df = pd.DataFrame({k: [0] for k in c['columns']})
        rename_cols(df, c['outcome'], **c['kwargs'])
        assert list(df.columns) == c['result'], c
