"""The tile-prediction table: the data contract between the inference path and
``biscuit.threshold``.

On disk the reference reads ``tile_predictions_eval.csv`` / ``tile_predictions_val_epoch1.csv``
(or ``.parquet.gzip``) with Slideflow's headers ``{outcome}-y_true0`` (or ``-y_true``),
``{outcome}-y_pred0/1``, ``{outcome}-uncertainty0/1`` -- dash or underscore -- and renames
them to ``y_true, y_pred, uncertainty`` (``biscuit/utils.py:19-53``,
``experiment.py:688-699,982-988``).  ``y_pred`` is the class-1 mean, ``uncertainty`` the
class-1 standard deviation over the MC passes.
"""
import os

import numpy as np
import pandas as pd

from .errors import PredsContainNaNError

EVAL_NAME = 'tile_predictions_eval.csv'            # experiment.py:690
VAL_NAME = 'tile_predictions_val_epoch1.csv'       # experiment.py:926, utils.py:216


# The column contract as a table (the data of ``biscuit/utils.py:19-53``): consumer name -> Slideflow header
# suffixes in order of preference.  Every suffix is looked up with an underscore separator first, then with a
# dash (``utils.py:34-49`` checks ``underscore=True in df.columns``); a missing column falls through to the
# dash form of the LAST candidate, which is what the reference ends up asking pandas to rename.
CONTRACT = {
    'y_true': ('y_true0', 'y_true'),        # utils.py:22-23, fallback utils.py:38-39
    'y_pred': ('y_pred1',),                 # utils.py:26-27: the class-1 mean over the MC passes
    'uncertainty': ('uncertainty1',),       # utils.py:19-20: the class-1 standard deviation
}


def _header(name, outcome, underscore=False):
    return f"{outcome}{'_' if underscore else '-'}{CONTRACT[name][0]}"


def uncertainty_header(outcome, underscore=False):
    return _header('uncertainty', outcome, underscore)


def y_true_header(outcome, underscore=False):
    return _header('y_true', outcome, underscore)


def y_pred_header(outcome, underscore=False):
    return _header('y_pred', outcome, underscore)


def resolve_cols(columns, outcome, **given):
    """{header present in ``columns``: consumer name} for the three contract columns; ``given`` overrides a lookup."""
    have = set(columns)
    found = {}
    for name, suffixes in CONTRACT.items():
        pick = given.get(name)
        if pick is None:
            first = suffixes[0]
            pick = f'{outcome}_{first}' if f'{outcome}_{first}' in have else f'{outcome}-{first}'
            for alt in suffixes[1:]:                      # only the dash form of a fallback exists in the reference
                if pick not in have:
                    pick = f'{outcome}-{alt}'
        found[pick] = name
    return found


def rename_cols(df, outcome, *, y_true=None, y_pred=None, uncertainty=None):
    """In-place rename to ``y_true / y_pred / uncertainty``; same accepted headers and the same result as
    ``biscuit/utils.py:31-53`` (a header that is absent is simply not renamed, as with ``DataFrame.rename``)."""
    df.rename(columns=resolve_cols(df.columns, outcome, y_true=y_true, y_pred=y_pred, uncertainty=uncertainty), inplace=True)


def tile_frame(outcome, slides, y_true, mean2, std2, loc=None):
    """Assemble the Slideflow-style tile table from device results.

    slides: sequence of slide names per tile; y_true: int per tile; mean2/std2: [T,2]."""
    mean2 = np.asarray(mean2, dtype=np.float32)
    std2 = np.asarray(std2, dtype=np.float32)
    if np.isnan(mean2).any():
        raise PredsContainNaNError('MC-dropout means contain NaN (threshold.py:141-142 would reject them)')
    cols = {'slide': pd.Series(list(slides), dtype=str)}
    if loc is not None:
        cols['loc_x'] = np.asarray(loc)[:, 0]
        cols['loc_y'] = np.asarray(loc)[:, 1]
    cols[f'{outcome}-y_true0'] = np.asarray(y_true).astype(np.int64)
    for k in (0, 1):
        cols[f'{outcome}-y_pred{k}'] = mean2[:, k].astype(np.float64)
    for k in (0, 1):
        cols[f'{outcome}-uncertainty{k}'] = std2[:, k].astype(np.float64)
    return pd.DataFrame(cols)


def save_tile_predictions(df, directory, name=EVAL_NAME):
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, name)
    if name.endswith('.parquet.gzip'):
        df.to_parquet(path, compression='gzip')
    else:
        df.to_csv(path, index=False)
    return path


def load_tile_predictions(path, outcome):
    """Read a tile table the way ``experiment.py:688-699`` does (slide as str) and rename."""
    if path.endswith('.parquet.gzip') or path.endswith('.parquet'):
        df = pd.read_parquet(path)
        df['slide'] = df['slide'].astype(str)
    else:
        df = pd.read_csv(path, dtype={'slide': str})
    rename_cols(df, outcome)
    return df
