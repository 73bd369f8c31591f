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
    """The pandas writer (``DataFrame.to_csv(index=False)`` / ``to_parquet``), as Slideflow writes the table: serial, after the
    run.  ``TableWriter`` below writes the same CSV bytes while the GPU works; this one stays as its checker and for parquet."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, name)
    if name.endswith('.parquet.gzip'):
        df.to_parquet(path, compression='gzip')
    else:
        df.to_csv(path, index=False)
    return path


class TableWriter:
    """The native writer of the tile table (``bqio_table_*`` of libbiscuit_io, ``csrc/table_writer.cpp``): rows are appended one
    run of tiles of one slide at a time, in the bytes ``DataFrame.to_csv(index=False)`` would write for ``tile_frame``'s columns
    (float64 cells as repr(float), the shortest string that reads back to the same double).  Raises ``PredsContainNaNError`` for
    a NaN prediction, as ``tile_frame`` does."""

    def __init__(self, path, outcome, with_loc=False, append=False):
        from . import tfrecord_native as tn
        self._lib = tn.lib()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.path = path
        self._h = self._lib.bqio_table_open(os.fsencode(path), str(outcome).encode(), int(bool(with_loc)), int(bool(append)))
        if not self._h:
            raise IOError(self._lib.bqio_table_last_error(None).decode())
        self.with_loc = bool(with_loc)

    def _check(self, e):
        if e == 0:
            return
        msg = self._lib.bqio_table_last_error(self._h).decode()
        if e == -6:                                              # BQIO_ERR_NAN
            raise PredsContainNaNError(msg + ' (threshold.py:141-142 would reject them)')
        raise (IOError if e == -2 else ValueError)(f'tile table {self.path}: {msg or e}')

    def rows(self, slide, y_true, mean2, std2, loc=None):
        mean2 = np.ascontiguousarray(mean2, dtype=np.float32)
        std2 = np.ascontiguousarray(std2, dtype=np.float32)
        n = mean2.shape[0]
        assert mean2.shape == (n, 2) and std2.shape == (n, 2)
        lp = None
        if loc is not None:
            loc = np.ascontiguousarray(loc, dtype=np.int64)
            assert loc.shape == (n, 2)
            lp = loc.ctypes.data
        self._check(self._lib.bqio_table_rows(self._h, str(slide).encode(), int(y_true), lp, mean2.ctypes.data, std2.ctypes.data, n))

    def tell(self):
        return int(self._lib.bqio_table_tell(self._h))

    def append_file(self, src, offset, length):
        self._check(self._lib.bqio_table_append_file(self._h, os.fsencode(src), int(offset), int(length)))

    def close(self):
        """(rows written by ``rows``, bytes written)."""
        import ctypes as C
        if not self._h:
            return 0, 0
        r, b = C.c_int64(0), C.c_int64(0)
        h, self._h = self._h, None
        e = self._lib.bqio_table_close(h, C.byref(r), C.byref(b))
        if e:
            raise IOError(f'tile table {self.path}: {self._lib.bqio_table_last_error(None).decode()}')
        return int(r.value), int(b.value)

    def __del__(self):
        try:
            self.close()
        except Exception:                                        # noqa: BLE001
            pass


def format_f64(v):
    """repr(float) by the native writer's formatter (for tests)."""
    import ctypes as C
    from . import tfrecord_native as tn
    buf = C.create_string_buffer(48)
    n = tn.lib().bqio_format_f64(float(v), buf, 48)
    assert n >= 0
    return buf.value.decode()


def write_tile_table(df, directory, name=EVAL_NAME, outcome=None):
    """A ``tile_frame`` DataFrame through the native writer (rows grouped into runs of one slide): the file
    ``save_tile_predictions`` would write, byte for byte.  For frames that are already in memory; ``evaluate`` streams instead."""
    cols = list(df.columns)
    with_loc = 'loc_x' in cols
    yt = next(c for c in cols if c.endswith('-y_true0'))
    outcome = yt[:-len('-y_true0')] if outcome is None else outcome
    path = os.path.join(directory, name)
    w = TableWriter(path, outcome, with_loc)
    slides = df['slide'].to_numpy()
    mean = np.stack([df[f'{outcome}-y_pred0'].to_numpy(), df[f'{outcome}-y_pred1'].to_numpy()], 1).astype(np.float32)
    std = np.stack([df[f'{outcome}-uncertainty0'].to_numpy(), df[f'{outcome}-uncertainty1'].to_numpy()], 1).astype(np.float32)
    y = df[yt].to_numpy()
    loc = np.stack([df['loc_x'].to_numpy(), df['loc_y'].to_numpy()], 1) if with_loc else None
    n = len(df)
    brk = np.flatnonzero((slides[1:] != slides[:-1]) | (y[1:] != y[:-1])) + 1 if n else np.zeros(0, np.int64)
    for a, b in zip(np.concatenate([[0], brk]).astype(int), np.concatenate([brk, [n]]).astype(int)):
        if b > a:
            w.rows(slides[a], y[a], mean[a:b], std[a:b], None if loc is None else loc[a:b])
    w.close()
    return path


# ---- per-rank shards of a multi-rank run -----------------------------------------------------------------------------------
# Rank r of W writes `{stem}.rank{r}.csv` and, next to it, `{stem}.rank{r}.csv.idx.json`: which slides (global index in dataset
# order) lie at which byte range of the shard.  After the run's one collective -- every rank closes its shard BEFORE it enters
# the all-gather, so leaving the gather means all shards are complete -- rank 0 splices the byte ranges into THE table in dataset
# order: the file a single-rank run writes, byte for byte, which is the file the reference's consumers open
# (experiment.py:688-699; threshold.detect takes Youden's J over EVERY tile of the cohort, threshold.py:417-426).

def shard_name(name, rank):
    stem, ext = (name[:-len('.parquet.gzip')], '.parquet.gzip') if name.endswith('.parquet.gzip') else os.path.splitext(name)
    return f'{stem}.rank{rank}{ext}'


def write_shard_index(path, rank, world, outcome, with_loc, slides):
    """slides: [[global slide index, name, rows, byte offset, byte length], ...] in the order they were written."""
    import json
    tmp = path + '.idx.json.tmp'
    with open(tmp, 'w') as f:
        json.dump({'rank': int(rank), 'world': int(world), 'outcome': outcome, 'with_loc': bool(with_loc),
                   'slides': [[int(a), str(b), int(c), int(d), int(e)] for a, b, c, d, e in slides]}, f)
    os.replace(tmp, path + '.idx.json')                          # (complete or absent, never half a file)


def remove_stale_shards(directory, world, name=EVAL_NAME):
    """Shards (and their indexes) of ranks >= ``world`` left in ``directory`` by an earlier run of a larger world: rank 0 removes them
    before it writes (ranks only ever write their own files, so this races with nobody); the splice would refuse the mixture."""
    import glob
    import re
    for path in glob.glob(os.path.join(directory, shard_name(name, '*'))) + glob.glob(os.path.join(directory, shard_name(name, '*') + '.idx.json')):
        m = re.search(r'\.rank(\d+)\.', os.path.basename(path))
        if m and int(m.group(1)) >= int(world):
            os.remove(path)


def find_shards(directory, name=EVAL_NAME):
    """[(shard path, its index dict)] sorted by rank, or [] -- raises when the set is not one complete world."""
    import glob
    import json
    found = []
    for idx in glob.glob(os.path.join(directory, shard_name(name, '*') + '.idx.json')):
        with open(idx) as f:
            found.append((idx[:-len('.idx.json')], json.load(f)))
    if not found:
        return []
    found.sort(key=lambda x: x[1]['rank'])
    world = found[0][1]['world']
    if [m['rank'] for _, m in found] != list(range(world)) or any(m['world'] != world for _, m in found):
        raise IOError(f'{directory}: shards of {name} are not one complete run (ranks {[m["rank"] for _, m in found]} of {world})')
    return found


def assemble_shards(directory, name=EVAL_NAME, remove=False):
    """Splice the ranks' shards into ``{directory}/{name}`` in dataset order (byte copies, nothing parsed).  Returns the path."""
    shards = find_shards(directory, name)
    if not shards:
        raise FileNotFoundError(f'{directory}: no shards of {name}')
    meta = shards[0][1]
    pieces = sorted((sl[0], path, sl[3], sl[4]) for path, m in shards for sl in m['slides'])
    if any(sl[2] and not sl[4] for _, m in shards for sl in m['slides']):
        raise IOError(f'{directory}: shards of {name} carry no byte ranges (written by pandas): load_tile_predictions(directory) reads them')
    if len({p[0] for p in pieces}) != len(pieces):
        raise IOError(f'{directory}: a slide appears in two shards of {name}')
    out = os.path.join(directory, name)
    w = TableWriter(out + '.tmp', meta['outcome'], meta['with_loc'])
    # neighbouring slides of one shard are neighbouring byte ranges (config 3: 200 slides of a rank = ONE copy)
    runs = []
    for _, path, off, ln in pieces:
        if runs and runs[-1][0] == path and runs[-1][1] + runs[-1][2] == off:
            runs[-1][2] += ln
        else:
            runs.append([path, off, ln])
    for path, off, ln in runs:
        w.append_file(path, off, ln)
    w.close()
    os.replace(out + '.tmp', out)
    if remove:
        for path, _ in shards:
            os.remove(path)
            os.remove(path + '.idx.json')
    return out


def _read_table(path):
    if path.endswith('.parquet.gzip') or path.endswith('.parquet'):
        df = pd.read_parquet(path)
        df['slide'] = df['slide'].astype(str)
        return df
    return pd.read_csv(path, dtype={'slide': str})               # experiment.py:692


def load_tile_predictions(path, outcome, name=EVAL_NAME):
    """Read a tile table the way ``experiment.py:688-699`` does (slide as str) and rename.  ``path``: the file, or the directory
    of a run -- its ``name`` when that exists, otherwise the ranks' shards put together in dataset order (in memory; CSV shards
    by their index files, parquet shards -- written whole by pandas -- by the order their slides appear in the index)."""
    if os.path.isdir(path):
        whole = os.path.join(path, name)
        if os.path.exists(whole):
            df = _read_table(whole)
        else:
            shards = find_shards(path, name)
            if not shards:
                raise FileNotFoundError(f'{path}: neither {name} nor its per-rank shards')
            parts = []
            for sp, m in shards:
                d = _read_table(sp)
                at = 0
                for gi, _, rows, _, _ in m['slides']:
                    parts.append((gi, d.iloc[at:at + rows]))
                    at += rows
                if at != len(d):
                    raise IOError(f'{sp}: {len(d)} rows, its index says {at}')
            parts.sort(key=lambda x: x[0])
            df = pd.concat([p for _, p in parts], ignore_index=True) if parts else _read_table(shards[0][0])
    else:
        df = _read_table(path)
    rename_cols(df, outcome)
    return df
