"""N>1 path on CPU: world_size-2 gloo processes run the real sharding / streaming /
gather code of biscuit_amd.inference with a stand-in for the device engine (the engine is
the only GPU-bound piece; its stand-in computes a deterministic function of tile content
and GLOBAL tile index so mis-sharding or mis-indexing changes the answer)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from biscuit_amd import distributed as D
from biscuit_amd.hp import nature2022
from biscuit_amd.inference import Slide, evaluate


class StandInEngine:
    def __init__(self):
        self.hp = nature2022()
        self.device = torch.device('cpu')
        self.calls = []

    def mc_infer(self, tiles, mc_n, seed, tile_idx0=0, mc_mode='head', out=None, tile_idx=None):
        n = tiles.shape[0]
        # (a batch across slide boundaries brings its tiles' global indices as an array: Engine.mc_infer(tile_idx=...))
        g = torch.arange(tile_idx0, tile_idx0 + n, dtype=torch.float64) if tile_idx is None else tile_idx.double() + tile_idx0
        p1 = tiles.reshape(n, -1).double().mean(1) / 255.0
        unc = ((g * 0.6180339887) % 1.0) * 0.05 + 0.001 * mc_n
        mean, std = out
        mean[:, 1] = p1.float(); mean[:, 0] = 1 - p1.float()
        std[:, 0] = unc.float(); std[:, 1] = unc.float()
        self.calls.append([int(x) for x in g.tolist()])           # the global indices this call covered
        return mean, std

    def slide_reduce(self, mean2, std2, slide_idx, n_slides, tile_uq=None, acc=None):
        if acc is None:
            acc = (torch.zeros(n_slides, dtype=torch.float64), torch.zeros(n_slides, dtype=torch.float64),
                   torch.zeros(n_slides, dtype=torch.int32))
        keep = torch.ones(mean2.shape[0], dtype=torch.bool) if not tile_uq else (std2[:, 1] < tile_uq)
        idx = slide_idx.long()[keep]
        acc[0].index_add_(0, idx, mean2[keep, 1].double())
        acc[1].index_add_(0, idx, std2[keep, 1].double())
        acc[2].index_add_(0, idx, torch.ones(idx.shape[0], dtype=torch.int32))
        return acc

    def slide_finish(self, acc):
        c = acc[2].double()
        return acc[0] / c, acc[1] / c, acc[2]


class StandInPool:
    """Same surface as biscuit_amd.engine.EnginePool, on CPU."""
    def __init__(self, n):
        self.engines = [StandInEngine() for _ in range(n)]
        self.hp, self.device = self.engines[0].hp, self.engines[0].device

    def __len__(self):
        return len(self.engines)

    def run(self, i, fn, wait_for_current=False):
        return fn(self.engines[i % len(self.engines)])

    def synchronize(self):
        pass


def make_slides(counts=None):
    rng = np.random.default_rng(0)
    counts = [7, 0, 13, 5, 21, 1, 9, 4] if counts is None else list(counts)
    return [Slide(f's{i}', rng.integers(0, 256, (c, 4, 4, 3), dtype=np.uint8), c, y_true=i % 2)
            for i, c in enumerate(counts)]


def reference_result(slides, tile_uq=None):
    pred, unc, cnt = [], [], []
    off = 0
    for s in slides:
        g = np.arange(off, off + s.n_tiles, dtype=np.float64)
        off += s.n_tiles
        p = s.tiles.reshape(s.n_tiles, -1).astype(np.float64).mean(1) / 255.0 if s.n_tiles else np.zeros(0)
        u = (((g * 0.6180339887) % 1.0) * 0.05 + 0.001 * 30).astype(np.float32).astype(np.float64)
        p = p.astype(np.float32).astype(np.float64)
        keep = np.ones(len(p), bool) if not tile_uq else u < tile_uq
        pred.append(p[keep].mean() if keep.any() else np.nan)
        unc.append(u[keep].mean() if keep.any() else np.nan)
        cnt.append(int(keep.sum()))
    return np.array(pred), np.array(unc), np.array(cnt)


@pytest.mark.parametrize('batch', [4, 16, 256])
def test_streaming_single_process(batch):
    slides = make_slides()
    eng = StandInEngine()
    res = evaluate(eng, slides, outcome='cohort', mc_n=30, seed=1, batch=batch)
    pred, unc, cnt = reference_result(slides)
    np.testing.assert_allclose(res.slide_pred, pred, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(res.slide_unc, unc, atol=1e-12, equal_nan=True)
    assert list(res.slide_count) == list(cnt)
    assert len(res.tile_df) == sum(cnt)
    assert list(res.tile_df.columns)[:2] == ['slide', 'cohort-y_true0']
    # one mc_infer call per batch (a batch across slides brings its indices as an array), all tiles exactly once
    seen = sorted(sum(eng.calls, []))
    assert seen == list(range(sum(cnt)))
    # slide table in first-appearance order, empty slide dropped
    sf, _ = res.slide_frame()
    assert list(sf['slide']) == [s.name for s in slides if s.n_tiles]


def test_engine_pool_round_robin():
    slides = make_slides()
    pool = StandInPool(2)
    res = evaluate(pool, slides, mc_n=30, seed=1, batch=8)
    pred, unc, cnt = reference_result(slides)
    np.testing.assert_allclose(res.slide_pred, pred, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(res.slide_unc, unc, atol=1e-12, equal_nan=True)
    assert list(res.slide_count) == list(cnt) and len(res.tile_df) == sum(cnt)
    assert all(len(e.calls) > 0 for e in pool.engines)          # both contexts were used


def test_tile_uq_filter_on_reduce():
    slides = make_slides()
    res = evaluate(StandInEngine(), slides, mc_n=30, seed=1, batch=8, tile_uq=0.05)
    pred, unc, cnt = reference_result(slides, tile_uq=0.05)
    assert list(res.slide_count) == list(cnt)
    ok = cnt > 0
    np.testing.assert_allclose(res.slide_pred[ok], pred[ok], atol=1e-12)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(device_type='cpu')
    slides = make_slides()
    res = evaluate(StandInEngine(), slides, mc_n=30, seed=1, batch=8, rank=r, world=w)
    q.put((rank, res.slide_pred, res.slide_unc, res.slide_count, res.local_slides, len(res.tile_df)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    slides = make_slides()
    pred, unc, cnt = reference_result(slides)
    locals_ = []
    for rank, gp, gu, gc, loc, ntile in out:
        np.testing.assert_allclose(gp, pred, atol=1e-12, equal_nan=True)      # every rank holds all slides
        np.testing.assert_allclose(gu, unc, atol=1e-12, equal_nan=True)
        assert list(gc) == list(cnt)
        assert ntile == sum(cnt[i] for i in loc)                              # tile rows stay rank-local
        locals_ += loc
    assert sorted(locals_) == list(range(len(slides)))                        # disjoint cover


# ---- eight ranks (BASELINE config 3's world size), ragged tile counts, fewer slides than ranks --------------------------
def _worker_n(rank, world, port, q, counts, batch):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(device_type='cpu')
    slides = make_slides(counts)
    eng = StandInEngine()
    res = evaluate(eng, slides, mc_n=30, seed=1, batch=batch, rank=r, world=w)
    q.put((rank, res.slide_pred, res.slide_unc, res.slide_count, res.local_slides, len(res.tile_df), eng.calls))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, counts, batch):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, q, counts, batch)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize('counts,batch', [
    ([40] * 64, 16),                                    # config 3 in small: equal slides -> contiguous blocks of 8 per rank
    ([17, 3, 0, 41, 8, 8, 8, 29, 1, 0, 12, 5, 33, 2, 19, 7, 7, 40, 11, 6, 0, 23, 9], 16),   # ragged, with empty slides
    ([5, 9, 2], 4),                                     # fewer slides than ranks: five ranks hold nothing
])
def test_eight_rank_gloo_gather(counts, batch):
    world = 8
    out = _run_world(world, counts, batch)
    slides = make_slides(counts)
    pred, unc, cnt = reference_result(slides)
    parts = D.partition_slides(counts, world)
    covered = []
    for rank, gp, gu, gc, loc, ntile, calls in out:
        np.testing.assert_allclose(gp, pred, atol=1e-12, equal_nan=True)      # every rank holds every slide's result
        np.testing.assert_allclose(gu, unc, atol=1e-12, equal_nan=True)
        assert list(gc) == list(cnt)
        assert loc == parts[rank] and ntile == sum(cnt[i] for i in loc)       # the partition every rank derived by itself
        covered += loc
        # every call covered a run of consecutive GLOBAL tile indices; all of the rank's tiles exactly once
        off = D.global_tile_offsets(counts)
        want = sorted(sum([list(range(int(off[i]), int(off[i]) + counts[i])) for i in loc], []))
        assert sorted(sum(calls, [])) == want
    assert sorted(covered) == list(range(len(counts)))
    if len(set(counts)) == 1:
        # equal slides: contiguous blocks, so a batch that spans slides is ONE run of indices --
        # one head call per batch, never one per slide
        per = len(counts) // world
        assert [p for p in parts] == [list(range(r * per, (r + 1) * per)) for r in range(world)]
        for rank, *_, calls in out:
            assert len(calls) == -(-per * counts[0] // batch)


def test_partition_keeps_lpt_loads_and_hands_out_ties_in_order():
    rng = np.random.default_rng(3)
    for world in (2, 3, 8):
        for _ in range(20):
            counts = rng.integers(0, 50, rng.integers(1, 60)).tolist()
            parts = D.partition_slides(counts, world)
            assert sorted(sum(parts, [])) == list(range(len(counts)))
            # the loads of plain greedy LPT
            load = [0] * world
            for c in sorted(counts, reverse=True):
                load[load.index(min(load))] += c
            assert sorted(sum(counts[i] for i in p) for p in parts) == sorted(load)
            # slides of equal count are handed out in index order: rank r's are all below rank r+1's
            for c in set(counts):
                owners = [r for r in range(world) for i in parts[r] if counts[i] == c]
                ids = [i for r in range(world) for i in parts[r] if counts[i] == c]
                assert ids == sorted(ids) and owners == sorted(owners)


# ---- the launcher of `bench.py --gpus N` and the host resource plan of N ranks on one node (no GPU involved) ----------------------
def _bench(*argv, timeout=180):
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *argv], env=env, cwd=root, capture_output=True, text=True,
                       timeout=timeout)
    return p.returncode, p.stdout, p.stderr, time.time() - t0


def test_launcher_eight_ranks_rendezvous_and_relay_rank0():
    rc, out, err, _ = _bench('--gpus', '8', '--selftest-exit', 'none')
    assert rc == 0, err[-2000:]
    assert '"selftest": "ok"' in out and '"world": 8' in out


def test_launcher_returns_within_seconds_when_a_rank_dies_before_the_rendezvous():
    """Rank 5 of 8 exits with code 7 before it joins: the other seven sit in the rendezvous (process-group timeout: 120 s; NCCL's
    default: ten minutes).  The parent polls all children, stops the rest and reports the code -- the 8-GPU command the driver runs
    cannot hang on a dead rank."""
    rc, out, err, dt = _bench('--gpus', '8', '--selftest-exit', '5:7')
    assert rc == 7 and dt < 60, (rc, dt, err[-2000:])
    assert 'exited with code 7' in err


def test_rank_cores_partition_the_node():
    # eight ranks, 128 allowed cores, no NUMA information: eight disjoint contiguous slices of 16
    parts = [D.rank_cores(r, 8, allowed=range(128)) for r in range(8)]
    assert [len(p) for p in parts] == [16] * 8 and sorted(sum(parts, [])) == list(range(128))
    # two NUMA nodes of 64 cores, four GPUs on each: rank 5 is the second of the four ranks next to node 1
    near = D.rank_cores(5, 8, allowed=range(128), numa_cpus=range(64, 128), numa_peers=(1, 4))
    assert near == list(range(80, 96))
    # a cgroup that left this process fewer cores than ranks: still never empty, never outside the mask
    few = [D.rank_cores(r, 8, allowed=[3, 9, 11]) for r in range(8)]
    assert all(len(p) == 1 and p[0] in (3, 9, 11) for p in few)
    # the NUMA node's cores are not in the mask: fall back to the whole mask
    assert D.rank_cores(1, 2, allowed=range(8), numa_cpus=range(64, 128), numa_peers=(0, 1)) == [4, 5, 6, 7]
    assert D._cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]


def test_decoder_threads_follow_the_affinity_mask():
    from biscuit_amd import tfrecord_native as tn
    if not hasattr(os, 'sched_setaffinity'):
        pytest.skip('no sched_setaffinity')
    before = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, sorted(before)[:2])
        assert tn.default_threads() == min(2, len(before))
    finally:
        os.sched_setaffinity(0, before)
    assert tn.default_threads() >= 1
