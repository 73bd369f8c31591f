"""Golden vectors for the column contract: the reference's own ``biscuit.utils.rename_cols`` (``utils.py:31-53``),
imported here under the stubs of ``make_consumer_golden.import_reference``, applied to frames with every header
spelling the function distinguishes.  TEST INFRASTRUCTURE ONLY.  The fixture holds the input column lists, the
keyword arguments and the column lists the reference leaves behind -- data, no reference source.

usage: python oracle/make_rename_golden.py      (needs /root/reference; writes tests/golden/rename_cols.json)
"""
import itertools
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle.make_consumer_golden import import_reference        # noqa: E402


def cases():
    out = []
    for outcome in ('cohort', 'LUAD-vs-LUSC', 'a_b'):
        seps = {'-': '-', '_': '_'}
        # every dash / underscore combination of the three headers, with and without the -y_true fallback spelling
        for st, sp, su in itertools.product(seps, seps, seps):
            for ytrue in ('y_true0', 'y_true', None):
                cols = ['slide', 'loc_x']
                if ytrue:
                    cols.append(f'{outcome}{st}{ytrue}')
                cols += [f'{outcome}{sp}y_pred0', f'{outcome}{sp}y_pred1', f'{outcome}{su}uncertainty0',
                         f'{outcome}{su}uncertainty1']
                out.append({'columns': cols, 'outcome': outcome, 'kwargs': {}})
        # both spellings present at once: the underscore one wins (utils.py:36,42,46)
        out.append({'columns': ['slide', f'{outcome}-y_true0', f'{outcome}_y_true0', f'{outcome}-y_pred1', f'{outcome}_y_pred1',
                                f'{outcome}-uncertainty1', f'{outcome}_uncertainty1'], 'outcome': outcome, 'kwargs': {}})
        # -y_true0 absent, both -y_true and _y_true present: only the dash form is a fallback (utils.py:38-39)
        out.append({'columns': ['slide', f'{outcome}_y_true', f'{outcome}-y_true', f'{outcome}-y_pred1', f'{outcome}-uncertainty1'],
                    'outcome': outcome, 'kwargs': {}})
        # explicit overrides
        out.append({'columns': ['slide', 'label', 'p1', 'sigma', f'{outcome}-y_pred1'], 'outcome': outcome,
                    'kwargs': {'y_true': 'label', 'y_pred': 'p1', 'uncertainty': 'sigma'}})
        out.append({'columns': ['slide', f'{outcome}-y_true0', 'p1', f'{outcome}-uncertainty1'], 'outcome': outcome,
                    'kwargs': {'y_pred': 'p1'}})
        # nothing to rename / another outcome's columns / already renamed
        out.append({'columns': ['slide', 'other-y_true0', 'other-y_pred1', 'other-uncertainty1'], 'outcome': outcome, 'kwargs': {}})
        out.append({'columns': ['slide', 'y_true', 'y_pred', 'uncertainty'], 'outcome': outcome, 'kwargs': {}})
    out.append({'columns': ['slide', '3-y_true0', '3-y_pred1', '3-uncertainty1'], 'outcome': 3, 'kwargs': {}})   # str(outcome)
    return out


def main():
    ref = import_reference('/root/reference')['utils']
    res = []
    for c in cases():
        df = pd.DataFrame({k: [0] for k in c['columns']})
        ref.rename_cols(df, c['outcome'], **c['kwargs'])
        res.append(dict(c, result=list(df.columns)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'rename_cols.json')
    json.dump({'source': 'biscuit/utils.py:31-53 (rename_cols), imported from /root/reference', 'pandas': pd.__version__,
               'cases': res}, open(path, 'w'), indent=0)
    print('wrote', path, len(res), 'cases')


if __name__ == '__main__':
    main()
