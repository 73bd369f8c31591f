"""bench.py -- tiles/sec of the MC-dropout tile-inference hot path on MI355X.

Workload (BASELINE.json config 2): synthetic slides of 1000 tiles (299x299x3 uint8,
resident in HBM before the timed region), Xception bf16 backbone + fp32 MC head, MC = 30,
batch = 256.  A "step" is one batch of 256 tiles through
    stage (K0) -> backbone (K1-K5) -> 30 Philox-dropout head passes + Welford (K6)
    -> slide-level segmented reduce (K7).
``value`` = tiles processed by all ranks / max-over-ranks wall time of exactly K steps
(barrier + synchronize on both sides; at N>1 the timed region ends with the single
all-gather of the per-slide results).

mc_mode 'head' (default, reported as ``value``): backbone once per tile, the 30 stochastic
passes run in the head -- bit-identical to 30 full passes because every dropout layer sits
behind the global pool and BN is in inference mode.  ``full_mode_value`` times the
reference's loop structure (30 complete forward passes) on the same kernels, and
``cpu_baseline`` times the fp32 CPU oracle in that same full structure on the host cores.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--mc 30] [--batch 256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from biscuit_amd import distributed as D      # noqa: E402
from biscuit_amd.engine import EnginePool     # noqa: E402
from biscuit_amd.weights import synthetic_weights  # noqa: E402

# Algorithmic work per tile (derivation SURVEY.md section 8d)
FLOP_PER_TILE_HEAD = 16.711e9 + 30 * 6.296e6      # backbone once + 30 head passes
FLOP_PER_TILE_FULL = 30 * (16.711e9 + 6.296e6)
BYTES_PER_TILE_BF16 = 90.2e6                      # layer-boundary bf16 bytes, fused dw+pw/BN/ReLU/pool+add
PEAK_HBM = 8.0e12                                 # B/s   (MI355X_MICROARCH.md: 8 TB/s spec)
PEAK_BF16 = 2.5e15                                # FLOP/s dense bf16 MFMA
PEAK_F32 = 157.3e12
TILES_PER_SLIDE = 1000


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:    # cgroup v2
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:    # cgroup v1
            quota = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = max(1, min(n, int(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(weights, mc_n, seed, batch=8, budget_s=20.0):
    """The CPU oracle in the reference's loop structure -- N complete forward passes per
    batch, then mean / population std -- on this host's cores.  Bounded sample: whole
    forward passes over one batch are timed until the budget is spent; tiles/s at MC=N is
    batch * passes / (N * seconds)."""
    from biscuit_amd.synthetic import make_tiles
    from oracle.xception_ref import XceptionOracle, standardize
    threads = usable_cores()
    torch.set_num_threads(threads)
    orc = XceptionOracle(weights)
    tiles = make_tiles(batch, seed=11)
    idx = np.arange(batch)
    x = standardize(tiles)
    orc.head_pass(orc.backbone(x), idx, 0, seed)              # warm-up (oneDNN primitive cache)
    passes, t0 = 0, time.time()
    while True:
        orc.head_pass(orc.backbone(x), idx, passes, seed)     # one full stochastic forward pass
        passes += 1
        dt_full = time.time() - t0
        if dt_full >= budget_s or passes >= mc_n:
            break
    full = batch * passes / (mc_n * dt_full)
    t0 = time.time()
    feat = orc.backbone(x)
    for p in range(mc_n):
        orc.head_pass(feat, idx, p, seed)
    dt_head = time.time() - t0
    return {'value': full, 'unit': 'tiles/s', 'cores': threads, 'kind': 'port',
            'sample': f'{passes} complete fp32 forward passes (PyTorch-CPU oracle, dropout on) over a batch of '
                      f'{batch} synthetic 299x299x3 tiles in {dt_full:.1f} s, scaled to MC={mc_n} passes per tile',
            'head_mode_value': batch / dt_head, 'cpu': _cpu_name(), 'os_cpu_count': os.cpu_count()}


def _cpu_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mc', type=int, default=30)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--mode', default='head', choices=['head', 'full'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--cpu-tiles', type=int, default=8)
    ap.add_argument('--streams', type=int, default=4,
                    help='batches in flight: independent contexts on HIP streams that own disjoint groups of XCDs (2 or 4)')
    args = ap.parse_args()

    rank, world, local = D.init_from_env('cuda')
    if world != args.gpus and rank == 0:
        print(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using {world}', file=sys.stderr)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    # collectives run on the GPU (RCCL) unless a CPU backend was forced for single-GPU testing
    coll_dev = dev if (world == 1 or dist.get_backend() == 'nccl') else torch.device('cpu')

    weights = synthetic_weights(1)
    pool_e = EnginePool(weights, n_streams=args.streams, dtype=args.dtype, max_batch=args.batch, max_mc=args.mc,
                        device=local)
    eng = pool_e.engines[0]
    NS = len(pool_e)
    B, K, Wm = args.batch, args.steps, args.warmup
    seed = 1234

    # synthetic tiles, generated on the device: 4 batches resident, cycled
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    pool = [torch.randint(0, 256, (B, 299, 299, 3), dtype=torch.uint8, device=dev, generator=g)
            for _ in range(4)]
    NSTEP = max(K, Wm, 8)                         # the per-kernel pass and the full-mode leg run up to 8 steps
    n_slides_local = (NSTEP * B + TILES_PER_SLIDE - 1) // TILES_PER_SLIDE + 1
    slide_of = [torch.div(torch.arange(s * B, (s + 1) * B, device=dev), TILES_PER_SLIDE,
                          rounding_mode='floor').to(torch.int32) for s in range(NSTEP)]
    mean = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(NS)]
    std = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(NS)]
    tile_base = rank * K * B                      # global tile index of this rank's shard

    def zero_acc():
        # one fixed-point accumulator triple per stream; integer sums add exactly at the end
        return [(torch.zeros(n_slides_local, dtype=torch.int64, device=dev),
                 torch.zeros(n_slides_local, dtype=torch.int64, device=dev),
                 torch.zeros(n_slides_local, dtype=torch.int32, device=dev)) for _ in range(NS)]

    def step(i, acc, mode):
        k = i % len(pool_e)           # batches in flight (set by the calibration below)

        def work(e):
            e.mc_infer(pool[i % 4], args.mc, seed, tile_idx0=tile_base + i * B, mc_mode=mode, out=(mean[k], std[k]))
            e.slide_reduce(mean[k], std[k], slide_of[i], n_slides_local, acc=acc[k])
        pool_e.run(k, work)

    def finish(acc):
        torch.cuda.synchronize()
        tot = tuple(sum(a[j] for a in acc[1:]) + acc[0][j] if NS > 1 else acc[0][j] for j in range(3))
        return eng.slide_finish(tot)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, steps):
        acc = zero_acc()
        for i in range(Wm):
            step(i, acc, mode)
        acc = zero_acc()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, acc, mode)
        mp, mu, cnt = finish(acc)
        if world > 1:                              # the path's one collective (RCCL over xGMI)
            ids = np.arange(rank * n_slides_local, (rank + 1) * n_slides_local)
            D.gather_slide_results(ids, mp.cpu().numpy(), mu.cpu().numpy(), cnt.cpu().numpy(),
                                   world * n_slides_local, n_slides_local, device=coll_dev)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert int(cnt.sum()) == steps * B and bool(torch.isfinite(mp[cnt > 0]).all())
        return dt

    # Untimed calibration of the number of batches in flight.  Each batch in flight owns 1/n of the chip's
    # XCDs (CU-masked streams), which lets batches run out of phase (one's HBM-bound prologues and store
    # drains under another's compute): 4 in flight measured 12.3 ms per batch, 2 12.5, one whole-chip stream
    # 13.0.  But n in flight only pays when the K timed steps fill whole rounds of n (10 steps on 4
    # quarter-chips are 3 rounds), and plain streams (if CU masks are unavailable) are bimodal.  So time
    # min(K, 32) steps each way and keep the fastest; every rank adopts the same choice.
    cands = sorted({n for n in (NS, NS // 2, 1) if n >= 1}, reverse=True)
    if len(cands) > 1:
        cal = min(K, 32)
        times = []
        for n in cands:
            pool_e.set_in_flight(n)
            times.append(timed(args.mode, cal))
        tt = torch.tensor(times, dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        pool_e.set_in_flight(cands[int(torch.argmin(tt).item())])
    streams_used = len(pool_e)
    dt = timed(args.mode, K)
    value = world * K * B / dt

    out = {
        'metric': 'tiles/sec at MC-dropout=30, 299x299x3 (Xception, slide-level pred/sigma reduce)',
        'value': value, 'unit': 'tiles/s', 'n_gpus': world, 'steps': K, 'warmup': Wm,
        'ms_per_step': dt / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': f'BASELINE.json config 2: {TILES_PER_SLIDE} synthetic 299x299x3 tiles/slide, '
                               f'Xception {args.dtype} + fp32 MC head, MC={args.mc}, batch={B}, '
                               f'{K * B} tiles/GPU resident in HBM',
                   'mc_mode': args.mode, 'mc_n': args.mc, 'batch': B, 'hip_streams': streams_used,
                   'parallelism': f'slide-sharded dp{world}, one all-gather of slide (pred, sigma, n)'},
    }

    if rank == 0:
        flop_tile = FLOP_PER_TILE_HEAD if args.mode == 'head' else FLOP_PER_TILE_FULL
        per_gpu = value / world
        out['path_roofline'] = {
            'hbm_frac': per_gpu * BYTES_PER_TILE_BF16 / PEAK_HBM if args.dtype == 'bf16' else None,
            'mfma_frac': per_gpu * flop_tile / (PEAK_BF16 if args.dtype == 'bf16' else PEAK_F32),
            'bytes_per_tile': BYTES_PER_TILE_BF16, 'flop_per_tile': flop_tile}

    # the other MC structure on the same kernels (N=1 only; a few steps)
    if world == 1 and rank == 0:
        other = 'full' if args.mode == 'head' else 'head'
        k2 = max(2, len(pool_e)) if other == 'full' else K     # a batch per stream keeps every XCD group busy
        dt2 = timed(other, k2)
        out[f'{other}_mode_value'] = k2 * B / dt2

    # per-kernel roofline: HIP events on the launch stream around every launch
    if not args.no_profile and rank == 0:
        acc = zero_acc()
        torch.cuda.synchronize()
        eng.profile_enable(True)
        psteps = 4
        # Engine 0 on a plain (whole-chip) stream of its own, nothing else in flight: the events bracket
        # each launch on the stream it runs on and give the kernel's own duration, the figure the
        # single-stream rocprofv3 trace under profiles/ is comparable with.  (In the timed region each
        # batch owns a quarter of the chip and four run side by side; see `in_situ` below.)
        solo = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(solo):
            for i in range(psteps):
                eng.mc_infer(pool[i % 4], args.mc, seed, tile_idx0=tile_base + i * B, mc_mode=args.mode,
                             out=(mean[0], std[0]))
                eng.slide_reduce(mean[0], std[0], slide_of[i], n_slides_local, acc=acc[0])
        solo.synchronize()
        ents = eng.profile_read()
        eng.profile_enable(False)
        tot = sum(e.ms for e in ents)
        ents.sort(key=lambda e: -e.ms)
        dom = ents[0]
        avg_s = dom.ms / dom.launches * 1e-3
        es = 2 if args.dtype == 'bf16' else 4
        ridge = (PEAK_BF16 if es == 2 else PEAK_F32) / PEAK_HBM
        bound = 'mfma' if dom.flops / max(dom.bytes, 1) >= ridge * 0.5 else 'hbm'
        if bound == 'mfma':
            peak = (PEAK_BF16 if es == 2 else PEAK_F32) / 1e12
            ach = dom.flops / avg_s / 1e12
            unit = 'TFLOP/s'
        else:
            peak = PEAK_HBM / 1e9
            ach = dom.bytes / avg_s / 1e9
            unit = 'GB/s'
        traffic = None      # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            if args.dtype == 'bf16' and B == 256 and dom.name in tj:
                traffic = tj[dom.name]['corrected_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            pass
        # the same kernel class inside the timed region: its work per step over its share of the step time
        in_situ = (dom.flops if bound == 'mfma' else dom.bytes) * (dom.launches / psteps) / \
                  ((dom.ms / tot) * (dt / K)) / (1e12 if bound == 'mfma' else 1e9)
        out['roofline'] = {'kernel': dom.name, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit,
                           'frac': ach / peak, 'traffic': traffic,
                           'in_situ': {'achieved': in_situ, 'frac': in_situ / peak,
                                       'note': 'work per step / (share of kernel time x measured step time), '
                                               f'{streams_used} batches in flight on disjoint XCD groups'},
                           'launches_per_step': dom.launches / psteps, 'avg_launch_ms': dom.ms / dom.launches,
                           'share_of_step': dom.ms / tot,
                           'algorithmic_flops_per_launch': dom.flops, 'algorithmic_bytes_per_launch': dom.bytes}
        out['kernels'] = [{'name': e.name, 'launches_per_step': e.launches / psteps,
                           'ms_per_launch': e.ms / e.launches, 'share': e.ms / tot,
                           'tflops': e.flops / (e.ms / e.launches * 1e-3) / 1e12,
                           'gbps': e.bytes / (e.ms / e.launches * 1e-3) / 1e9} for e in ents[:12]]

    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(weights, args.mc, seed, args.cpu_tiles)
        out['speedup_vs_cpu_full'] = (out.get('full_mode_value') or value) / out['cpu_baseline']['value']
        out['speedup_headline_vs_cpu_full'] = value / out['cpu_baseline']['value']

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
