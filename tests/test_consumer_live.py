"""Differential test of biscuit_amd.threshold against the reference's own biscuit/threshold.py,
imported live.  Only runs where /root/reference exists (the build container); on the GPU box the
committed golden fixtures (tests/test_consumer.py) carry the same pin."""
import os
import sys

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'biscuit')), reason='reference not mounted')


@pytest.fixture(scope='module')
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle.make_consumer_golden import import_reference
    return import_reference(REF)['threshold']


def _same(a, b):
    if a is None or b is None:
        assert a is None and b is None
    else:
        assert float(a) == float(b) or (np.isnan(float(a)) and np.isnan(float(b)))


@pytest.mark.parametrize('seed', range(12))
def test_random_frames_match_reference(ref, seed):
    from biscuit_amd import threshold as th
    from oracle.make_consumer_golden import make_frame
    rng = np.random.default_rng(1000 + seed)
    df = make_frame(seed=500 + seed, n_slides=int(rng.integers(6, 40)), tiles_per_slide=int(rng.integers(4, 50)),
                    flip_frac=float(rng.uniform(0.1, 0.45)), ragged=bool(seed % 2))
    patients = dict(zip(df['slide'], df['patient']))
    # detect
    try:
        want_thr, want_auc = ref.detect(df.copy(), patients=patients)
        want_err = None
    except Exception as e:  # noqa: BLE001
        want_err = type(e).__name__
    if want_err:
        with pytest.raises(Exception) as ei:
            th.detect(df.copy(), patients=patients)
        assert type(ei.value).__name__ == want_err
        return
    got_thr, got_auc = th.detect(df.copy(), patients=patients)
    for k in want_thr:
        _same(got_thr[k], want_thr[k])
    _same(got_auc, want_auc)
    # apply at both levels with data-derived thresholds
    tile_uq = float(np.quantile(df['uncertainty'], rng.uniform(0.3, 0.9)))
    for level in ('slide', 'patient'):
        slide_uq = float(rng.uniform(0.005, 0.04))
        w_res, w_df = ref.apply(df.copy(), tile_uq=tile_uq, slide_uq=slide_uq, tile_pred=0.5, slide_pred=0.45,
                                patients=patients, level=level)
        g_res, g_df = th.apply(df.copy(), tile_uq=tile_uq, slide_uq=slide_uq, tile_pred=0.5, slide_pred=0.45,
                               patients=patients, level=level)
        for k in w_res:
            _same(g_res[k], w_res[k])
        if w_df is None:
            assert g_df is None
        else:
            assert list(g_df[level]) == list(w_df[level])
            for c in ('y_pred', 'uncertainty', 'error'):
                np.testing.assert_allclose(g_df[c].to_numpy(), w_df[c].to_numpy(), atol=1e-12)
            for c in ('y_true', 'correct', 'incorrect', 'y_pred_bin'):
                assert [int(x) for x in g_df[c]] == [int(x) for x in w_df[c]]


def test_rename_cols_matches_reference_live():
    """Random column sets through both ``rename_cols`` (reference: utils.py:31-53)."""
    import pandas as pd
    from oracle.make_consumer_golden import import_reference
    from biscuit_amd.predictions import rename_cols
    ref_utils = import_reference(REF)['utils']
    rng = np.random.default_rng(7)
    names = ['y_true0', 'y_true', 'y_pred0', 'y_pred1', 'uncertainty0', 'uncertainty1']
    for _ in range(300):
        outcome = str(rng.choice(['cohort', 'x-y', 'a_b', '7']))
        cols = ['slide'] + [f'{outcome}{sep}{n}' for n in names for sep in '-_' if rng.random() < 0.45]
        cols += [c for c in ('y_true', 'y_pred', 'label') if rng.random() < 0.15]
        kw = {}
        if rng.random() < 0.2:
            kw['y_pred'] = str(rng.choice(cols))
        a = pd.DataFrame({c: [0] for c in cols}); b = a.copy()
        ref_utils.rename_cols(a, outcome, **kw)
        rename_cols(b, outcome, **kw)
        assert list(a.columns) == list(b.columns), (cols, outcome, kw)
