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


def uncertainty_header(outcome, underscore=False):
    return str(outcome) + ('_' if underscore else '-') + 'uncertainty1'


def y_true_header(outcome, underscore=False):
    return str(outcome) + ('_' if underscore else '-') + 'y_true0'


def y_pred_header(outcome, underscore=False):
    return str(outcome) + ('_' if underscore else '-') + 'y_pred1'


def rename_cols(df, outcome, *, y_true=None, y_pred=None, uncertainty=None):
    """In-place rename to ``y_true / y_pred / uncertainty`` (``utils.py:31-53``): accepts
    dash or underscore headers and the ``{outcome}-y_true`` fallback."""
    if y_true is None:
        y_true = y_true_header(outcome, underscore=(y_true_header(outcome, True) in df.columns))
        if y_true not in df.columns:
            y_true = str(outcome) + '-y_true'
    if y_pred is None:
        y_pred = y_pred_header(outcome, underscore=(y_pred_header(outcome, True) in df.columns))
    if uncertainty is None:
        uncertainty = uncertainty_header(outcome, underscore=(uncertainty_header(outcome, True) in df.columns))
    df.rename(columns={y_true: 'y_true', y_pred: 'y_pred', uncertainty: 'uncertainty'}, inplace=True)


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
