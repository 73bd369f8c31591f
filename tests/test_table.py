"""The tile-prediction table on disk -- the product ``biscuit.threshold`` reads (``experiment.py:688-699``; column contract
``utils.py:19-53``) -- written by the native writer of libbiscuit_io while the run is in flight, and put together from the
ranks' shards of a multi-rank run.  CPU only: the engine is the stand-in of ``tests/test_distributed.py``; the GPU counterpart
(two real ranks on one device) is in ``tests/test_gpu_configs.py``.

The checker is pandas: ``DataFrame.to_csv(index=False)`` of ``predictions.tile_frame`` is how Slideflow leaves the file."""
import hashlib
import os
import socket
import struct

import numpy as np
import pandas as pd
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from biscuit_amd import distributed as D, predictions as P, threshold
from biscuit_amd.errors import PredsContainNaNError
from biscuit_amd.inference import Slide, evaluate
from tests.test_distributed import StandInEngine, StandInPool


def _sha(path):
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def test_float_cells_are_python_repr_and_round_trip_bit_exactly():
    """Every float64 the writer emits is repr(float): the shortest string that reads back to the same double, in the layout pandas
    writes.  Specials, both layout boundaries (1e-4, 1e16), denormals, float32 values widened to float64 (what the table holds)."""
    fixed = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999999e-5, 1e-5, 1.5e-7, 1e15, 1e16, 9999999999999998.0, 123456789012345678.0, 5e-324,
             2.2250738585072014e-308, 1.7976931348623157e308, 0.30000001192092896, 12345.678, 1e22, 1e-310, 100.0, 65504.0]
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 64, 20000, dtype=np.uint64)
    rand = np.frombuffer(bits.tobytes(), dtype=np.float64)
    rand = rand[np.isfinite(rand)]
    f32 = rng.random(20000, dtype=np.float32).astype(np.float64) * (10.0 ** rng.integers(-8, 1, 20000))
    for v in list(fixed) + rand.tolist() + f32.tolist():
        s = P.format_f64(v)
        assert s == repr(float(v)), (v, s)
        assert struct.pack('<d', float(s)) == struct.pack('<d', float(v))
    assert P.format_f64(float('inf')) == 'inf' and P.format_f64(float('-inf')) == '-inf' and P.format_f64(float('nan')) == ''


def _frame(n=5000, with_loc=False, seed=0):
    rng = np.random.default_rng(seed)
    mean = rng.random((n, 2)).astype(np.float32)
    mean[:50] *= 1e-6
    mean[50:100] *= 1e-3
    mean[5] = 0
    mean[6, 0] = 1
    mean[7, 0] = 1e-5
    std = (rng.random((n, 2)) * 0.1).astype(np.float32)
    std[3, 0] = np.nan                         # (a NaN uncertainty is pandas' empty cell; only NaN predictions are refused)
    std[4, 1] = np.inf
    slides = [f's{i // 100:04d}' for i in range(n)]
    slides[0:100] = ['we,ird "name"\n'] * 100
    slides[100:200] = ['00123'] * 100          # all digits: must come back as str (experiment.py:692 reads slide as str)
    y = [(i // 100) % 2 for i in range(n)]
    loc = rng.integers(0, 100000, (n, 2)) if with_loc else None
    return P.tile_frame('cohort', slides, y, mean, std, loc)


@pytest.mark.parametrize('with_loc', [False, True])
def test_native_writer_is_the_pandas_writer_byte_for_byte(tmp_path, with_loc):
    df = _frame(with_loc=with_loc)
    a = P.save_tile_predictions(df, str(tmp_path), 'pandas.csv')
    b = P.write_tile_table(df, str(tmp_path), 'native.csv')
    assert open(a, 'rb').read() == open(b, 'rb').read()
    back = pd.read_csv(b, dtype={'slide': str})
    assert back.equals(pd.read_csv(a, dtype={'slide': str}))
    # every float64 round-trips bit-exactly (NaN as NaN) through an exact parser; pandas' DEFAULT parser -- the one the reference's
    # `pd.read_csv(path, dtype={'slide': str})` uses -- is its fast one, off by an ulp on some cells, identically for both files
    back = pd.read_csv(b, dtype={'slide': str}, float_precision='round_trip')
    for col in df.columns:
        if df[col].dtype == np.float64:
            assert np.array_equal(back[col].to_numpy().view(np.uint64), df[col].to_numpy().view(np.uint64)), col
    # the consumer's reader renames by the contract and sees the same table
    got = P.load_tile_predictions(b, 'cohort')
    assert {'y_true', 'y_pred', 'uncertainty', 'slide'} <= set(got.columns) and got['slide'].iloc[150] == '00123'


def test_nan_prediction_is_refused_and_locations_must_match(tmp_path):
    w = P.TableWriter(str(tmp_path / 't.csv'), 'cohort')
    ok = np.full((3, 2), 0.5, np.float32)
    bad = ok.copy()
    bad[1, 1] = np.nan
    w.rows('a', 0, ok, ok)
    with pytest.raises(PredsContainNaNError):
        w.rows('b', 1, bad, ok)
    with pytest.raises(ValueError):
        w.rows('c', 1, ok, ok, loc=np.zeros((3, 2), np.int64))
    assert w.close()[0] == 3
    assert len(pd.read_csv(tmp_path / 't.csv')) == 3               # nothing of the refused calls was written
    with pytest.raises(IOError):
        P.TableWriter(str(tmp_path / 'no' / 'such' / '\0bad'), 'cohort')


def _slides(counts, with_loc=False, seed=0):
    rng = np.random.default_rng(seed)
    return [Slide(f's{i}', rng.integers(0, 256, (c, 4, 4, 3), dtype=np.uint8), c, y_true=i % 2,
                  loc=(rng.integers(0, 9999, (c, 2)) if with_loc else None)) for i, c in enumerate(counts)]


COUNTS = [17, 3, 0, 41, 8, 8, 8, 29, 1, 0, 12, 5, 33, 2, 19, 7]


@pytest.mark.parametrize('batch,with_loc,pool', [(4, False, False), (16, True, False), (256, False, False), (8, True, True)])
def test_evaluate_streams_the_table_it_returns(tmp_path, batch, with_loc, pool):
    """The file written while the run is in flight == ``to_csv`` of the frame the run returns, whatever the batch size (batches
    spanning several slides, a slide spanning several batches) -- and it is written without the frame too."""
    slides = _slides(COUNTS, with_loc)
    eng = StandInPool(2) if pool else StandInEngine()
    res = evaluate(eng, slides, outcome='cohort', mc_n=30, seed=1, batch=batch, save_dir=str(tmp_path / 'a'))
    assert res.table_path == str(tmp_path / 'a' / P.EVAL_NAME) and res.table_rows == sum(COUNTS) == len(res.tile_df)
    want = P.save_tile_predictions(res.tile_df, str(tmp_path / 'p'))
    assert open(res.table_path, 'rb').read() == open(want, 'rb').read()
    assert ('loc_x' in res.tile_df.columns) == with_loc
    lean = evaluate(StandInEngine(), slides, outcome='cohort', mc_n=30, seed=1, batch=batch, save_dir=str(tmp_path / 'b'),
                    keep_tiles=False, table_name=P.VAL_NAME)
    assert lean.tile_df is None and lean.table_path.endswith(P.VAL_NAME) and _sha(lean.table_path) == _sha(want)
    old = evaluate(StandInEngine(), slides, outcome='cohort', mc_n=30, seed=1, batch=batch, save_dir=str(tmp_path / 'c'),
                   table_writer='pandas')
    assert _sha(old.table_path) == _sha(want)


def test_a_nan_from_the_engine_fails_the_run(tmp_path):
    class NanEngine(StandInEngine):
        def mc_infer(self, tiles, *a, out=None, **k):
            m, s = super().mc_infer(tiles, *a, out=out, **k)
            if len(self.calls) == 3:
                m[0, 1] = float('nan')
            return m, s
    with pytest.raises(PredsContainNaNError):
        evaluate(NanEngine(), _slides(COUNTS), mc_n=30, seed=1, batch=8, save_dir=str(tmp_path), keep_tiles=False)


PARQUET = 'tile_predictions_eval.parquet.gzip'          # the other on-disk form the reference reads (utils.py:190-228)


def _rank(rank, world, port, counts, batch, out_dir, with_loc, writer):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(device_type='cpu')
    res = evaluate(StandInEngine(), _slides(counts, with_loc), outcome='cohort', mc_n=30, seed=1, batch=batch, save_dir=out_dir,
                   rank=r, world=w, keep_tiles=(writer != 'native'), table_writer='pandas' if writer == 'parquet' else writer,
                   table_name=PARQUET if writer == 'parquet' else P.EVAL_NAME)
    assert res.table_rows == sum(counts[i] for i in res.local_slides)
    if r == 0 and writer == 'native':
        assert res.table_path == os.path.join(out_dir, P.EVAL_NAME) and os.path.exists(res.table_path)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, counts, batch, out_dir, with_loc=False, writer='native'):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_rank, args=(r, world, port, counts, batch, out_dir, with_loc, writer)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0


@pytest.mark.parametrize('world,counts,with_loc', [
    (2, COUNTS, True),                          # ragged counts: LPT gives each rank a NON-contiguous set of slides
    (3, [5, 9, 2, 0, 7, 7, 7, 1], False),
    (8, [5, 9, 2], False),                      # fewer slides than ranks: five ranks write a header-only shard
])
def test_shards_of_a_multi_rank_run_make_the_single_rank_table(tmp_path, world, counts, with_loc):
    """world > 1: every rank streams its shard and closes it before the all-gather; rank 0 splices the shards -- byte ranges, no
    parsing -- into ONE table in dataset order: the single-rank file byte for byte, so ``threshold.detect`` (Youden over every
    tile of the cohort, threshold.py:417-426) finds the same thresholds at any world size."""
    multi, single = str(tmp_path / 'multi'), str(tmp_path / 'single')
    _run(world, counts, 8, multi, with_loc)
    one = evaluate(StandInEngine(), _slides(counts, with_loc), outcome='cohort', mc_n=30, seed=1, batch=8, save_dir=single)
    assert open(os.path.join(multi, P.EVAL_NAME), 'rb').read() == open(one.table_path, 'rb').read()
    shards = P.find_shards(multi)
    assert [m['rank'] for _, m in shards] == list(range(world))
    parts = D.partition_slides(counts, world)
    for (path, m), mine in zip(shards, parts):
        assert [s[0] for s in m['slides']] == [i for i in mine if counts[i]]
        assert len(pd.read_csv(path)) == sum(counts[i] for i in mine)
    # the reader put in front of a directory: THE table when it is there, the shards in dataset order when it is not
    a = P.load_tile_predictions(multi, 'cohort')
    os.remove(os.path.join(multi, P.EVAL_NAME))
    b = P.load_tile_predictions(multi, 'cohort')
    c = P.load_tile_predictions(single, 'cohort')
    assert a.equals(c) and b.equals(c) and list(c['slide'].unique()) == [f's{i}' for i, n in enumerate(counts) if n]
    if len(c['y_true'].unique()) == 2:
        t_multi, auc_multi = threshold.detect(b.copy())
        t_single, auc_single = threshold.detect(c.copy())
        assert t_multi == t_single and (auc_multi == auc_single or (np.isnan(auc_multi) and np.isnan(auc_single)))
    # splice again by hand, removing the shards
    again = P.assemble_shards(multi, remove=True)
    assert _sha(again) == _sha(one.table_path) and P.find_shards(multi) == []
    # an incomplete set of shards is an error, not a shorter table
    _run(world, counts, 8, str(tmp_path / 'gap'), with_loc)
    os.remove(os.path.join(tmp_path / 'gap', P.EVAL_NAME))
    os.remove(P.find_shards(str(tmp_path / 'gap'))[-1][0] + '.idx.json')
    with pytest.raises(IOError):
        P.load_tile_predictions(str(tmp_path / 'gap'), 'cohort')


def test_a_smaller_world_into_the_same_directory_replaces_the_larger_ones_shards(tmp_path):
    out = str(tmp_path / 'run')
    _run(3, COUNTS, 8, out)
    assert len(P.find_shards(out)) == 3
    _run(2, COUNTS, 8, out)                                # rank 0 removes rank 2's files of the earlier run before it writes
    assert [m['world'] for _, m in P.find_shards(out)] == [2, 2]
    one = evaluate(StandInEngine(), _slides(COUNTS), outcome='cohort', mc_n=30, seed=1, batch=8, save_dir=out)
    assert P.find_shards(out) == [] and _sha(one.table_path) == _sha(os.path.join(out, P.EVAL_NAME))


def test_pandas_shards_are_read_in_dataset_order(tmp_path):
    _run(2, COUNTS, 8, str(tmp_path / 'multi'), writer='pandas')
    one = evaluate(StandInEngine(), _slides(COUNTS), outcome='cohort', mc_n=30, seed=1, batch=8, save_dir=str(tmp_path / 'single'))
    assert P.load_tile_predictions(str(tmp_path / 'multi'), 'cohort').equals(P.load_tile_predictions(one.table_path, 'cohort'))
    with pytest.raises(IOError):
        P.assemble_shards(str(tmp_path / 'multi'))
    # the parquet form: shards written whole by pandas, read back in dataset order
    _run(2, COUNTS, 8, str(tmp_path / 'pq'), writer='parquet')
    onep = evaluate(StandInEngine(), _slides(COUNTS), outcome='cohort', mc_n=30, seed=1, batch=8, save_dir=str(tmp_path / 'pq1'),
                    table_name=PARQUET)
    assert onep.table_path.endswith(PARQUET)
    assert P.load_tile_predictions(str(tmp_path / 'pq'), 'cohort', name=PARQUET).equals(P.load_tile_predictions(onep.table_path, 'cohort'))


def test_cli_skips_a_finished_evaluation(tmp_path):
    """``--skip-existing``: the idempotence of the reference's Step 6 (``utils.eval_exists``, biscuit/experiment.py:913-914) -- a
    directory that already holds the table is left alone, before anything touches a GPU or a rendezvous (so this runs on CPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / P.EVAL_NAME).write_text('slide,cohort-y_true0\n')
    p = subprocess.run([sys.executable, '-m', 'biscuit_amd', '--synthetic', '2x2', '--out', str(tmp_path), '--skip-existing'], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])['skipped'] is True
    assert (tmp_path / P.EVAL_NAME).read_text() == 'slide,cohort-y_true0\n'
