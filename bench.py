"""bench.py -- tiles/sec of the MC-dropout tile-inference hot path on MI355X.

Default workload (BASELINE.json config 2): synthetic slides of 1000 tiles (299x299x3 uint8,
resident in HBM before the timed region), Xception with 16-bit storage and matrix cores + fp32 MC head, MC = 30,
batch = 256.  The headline dtype is f16 (IEEE half): the 16-bit mode that holds the north-star tolerance of 1e-3 on
tile and slide mean / sigma (tests/test_gpu_configs.py::test_hard_weights_throughput_mode_holds_tolerance); bf16, which
BASELINE config 2 names, runs 5-6 % slower (round 4) and misses that tolerance on O(1)-logit weights (2.7e-3); it is
reported next to it as ``bf16_value``.  A "step" is one batch of 256 tiles through
    stage (K0) -> backbone (K1-K5) -> 30 Philox-dropout head passes + Welford (K6)
    -> slide-level segmented reduce (K7).
``value`` = tiles processed by all ranks / max-over-ranks wall time of exactly K steps
(barrier + synchronize on both sides; at N>1 the timed region ends with the single
all-gather of the per-slide results).  Weak scaling: K x 256 tiles per GPU.

``--workload cfg3`` (BASELINE.json config 3, strong scaling): S = 1600 synthetic slides x T = 1000
tiles, longest-processing-time sharded over the ranks, streamed through the product's own
``biscuit_amd.inference.evaluate`` (batches of 256 that span slides, global Philox tile indices), one
all-gather of the slide (pred, sigma, n) table; ``value`` = S x T / max-over-ranks wall time.

mc_mode 'head' (default, reported as ``value``): backbone once per tile, the 30 stochastic
passes run in the head -- bit-identical to 30 full passes because every dropout layer sits
behind the global pool and BN is in inference mode.  Extra keys (N = 1 only, never the headline):
``full_mode_value`` (the reference's loop structure, 30 complete forward passes, same kernels),
``with_reinhard_value`` (hp.py:19's stain normaliser in the timed region), ``bf16_value`` / ``f32_value`` (the other
storage types on the same loop), ``b1_latency`` (``UncertaintyInterface`` on one tile, results.py:257), ``tfrecords``
(``evaluate`` from self-written PNG TFRecords: decode on the host cores) and ``cpu_baseline`` (the fp32
CPU oracle in the reference's loop structure on the host cores).

``--gpus N`` without torchrun's environment starts the N ranks itself (one child process per GPU,
before this process has touched the GPU); under ``python -m torch.distributed.run`` it is one of them.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--mc 30] [--batch 256] [--workload cfg2|cfg3]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic work per tile (derivation SURVEY.md section 8d)
FLOP_BACKBONE = 16.711e9
FLOP_HEAD_PASS = 6.296e6
BYTES_PER_TILE_BF16 = 90.2e6                      # layer-boundary 16-bit bytes, fused dw+pw/BN/ReLU/pool+add
PEAK_HBM = 8.0e12                                 # B/s   (MI355X_MICROARCH.md: 8 TB/s spec)
PEAK_HBM_MEASURED = 6.29e12                       # B/s   (same guide: measured copy rate)
PEAK_BF16 = 2.5e15                                # FLOP/s dense bf16 = f16 MFMA
HALF = ('f16', 'bf16')                            # the 16-bit storage / matrix-core types
PEAK_F32 = 157.3e12
TILES_PER_SLIDE = 1000
# profiling classes of the entry side (stage, stem, block1_conv2, blocks 2 and 3 with their fused ends): `entry_side_ms`
ENTRY_SIDE = ('stage_u8', 'stage_stats', 'front_stage_stem_conv2', 'stem_conv1', 'conv3x3_k32_n64_147', 'sepconv_k64_n128_147',
              'sepconv_k128_n128_147', 'respool_147', 'blocktail_147', 'sepconv_k128_n256_74', 'sepconv_k256_n256_74', 'respool_74',
              'blocktail_74')
NORM_FIT = {'target_means': [65.0, 12.0, -8.0], 'target_stds': [14.0, 7.0, 6.0]}   # a plausible H&E fit (synthetic)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200,
                    help='timed batches (2 s of work: the drain of the last batches in flight and the final reduce are\n                          inside the timed region, 3.5 %% of 20 steps)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mc', type=int, default=30)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'],
                    help='storage / matrix-core type of the backbone (f16: the mode that holds the 1e-3 tolerance)')
    ap.add_argument('--mode', default='head', choices=['head', 'full'])
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg3'])
    ap.add_argument('--slides', type=int, default=1600, help='cfg3: number of synthetic slides')
    ap.add_argument('--tiles-per-slide', type=int, default=TILES_PER_SLIDE, help='cfg3: tiles per slide')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ragged', action='store_true', help='cfg3: slides of unequal tile counts (T - (7 i mod T/2)): LPT shards are not contiguous')
    ap.add_argument('--no-table', action='store_true', help='cfg3: skip the second run that also writes the tile table')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the reinhard / f32 / B=1 / TFRecord legs')
    ap.add_argument('--cpu-tiles', type=int, default=128, help='batch of the CPU baseline (hp.py:7: 128)')
    ap.add_argument('--cpu-budget', type=float, default=25.0, help='seconds of CPU work in the baseline sample')
    ap.add_argument('--dist-backend', default=None, help='process-group backend (default: nccl = RCCL); tests: gloo')
    ap.add_argument('--local-device', type=int, default=None, help='GPU index of this rank (default: LOCAL_RANK)')
    ap.add_argument('--size-grids', action='store_true',
                    help='persistent grids sized for the CUs of each stream (default: for the whole chip)')
    ap.add_argument('--selftest-exit', default=None, metavar='R:CODE',
                    help='(tests, no GPU) rank R exits with CODE before the rendezvous ("none": nobody does), the other ranks run a '
                         'gloo rendezvous and a barrier: exercises the launcher of --gpus N')
    ap.add_argument('--fixed-streams', action='store_true', help='exactly --streams batches in flight, no calibration')
    ap.add_argument('--streams', type=int, default=4,
                    help='batches in flight: independent contexts on HIP streams that own disjoint groups of XCDs (2 or 4)')
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- rank launcher
def spawn_ranks(args, argv=None, poll_s=0.2, grace_s=5.0):
    """Start one child process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in its environment), relay rank 0's
    JSON line, return the worst exit code.  The parent never touches the GPU: nothing that has initialised HIP
    is ever exec'd or forked.  All children are polled: when one exits non-zero the others are terminated (a rank that dies
    before the rendezvous would otherwise leave them waiting for it -- and this parent waiting for them -- until the
    process-group timeout), and the parent returns that code within seconds."""
    import tempfile
    import threading
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        # ROCm's HIP IPC (RCCL's intra-node transport, CUDA-tensor sharing across processes) must use dmabuf handles on hosts whose
        # driver supports nothing else: with the legacy mode hipIpcGetMemHandle fails with "invalid argument".  The launcher
        # environment of this pool exports it already; children started from a clean environment get it here.
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv)),
                                      env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    worst, failed_at = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.time()
                worst = max(abs(c) for c in bad)
                for p, c in zip(procs, codes):
                    if c is None:
                        p.terminate()
            if all(c is not None for c in codes):
                break
            if failed_at is not None and time.time() - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if failed_at is None:
        worst = max(abs(p.returncode) for p in procs)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if failed_at is not None:
        print(f'[bench] a rank exited with code {worst}; the other ranks were stopped', file=sys.stderr)
    return worst


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


def _cpu_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(weights, mc_n, seed, batch=128, budget_s=25.0):
    """The CPU oracle in the reference's loop structure -- N complete forward passes per
    batch, then mean / population std -- on this host's cores, batch 128 (hp.py:7).  Bounded sample:
    whole forward passes over one batch are timed until the budget is spent (at least one, after a
    warm-up pass); tiles/s at MC=N is batch * passes / (N * seconds)."""
    import numpy as np
    import torch
    from biscuit_amd.synthetic import make_tiles
    from oracle.xception_ref import XceptionOracle, standardize
    threads = usable_cores()
    torch.set_num_threads(threads)
    orc = XceptionOracle(weights)
    tiles = np.concatenate([make_tiles(min(batch, 16), seed=11)] * ((batch + 15) // 16))[:batch]
    idx = np.arange(batch)
    x = standardize(tiles)
    orc.head_pass(orc.backbone(x[:min(batch, 16)]), idx[:min(batch, 16)], 0, seed)   # warm-up (oneDNN primitive cache)
    passes, t0 = 0, time.time()
    while True:
        orc.head_pass(orc.backbone(x), idx, passes, seed)     # one full stochastic forward pass
        passes += 1
        dt_full = time.time() - t0
        if dt_full >= budget_s or passes >= mc_n:
            break
    full = batch * passes / (mc_n * dt_full)
    t0 = time.time()
    feat = orc.backbone(x[:min(batch, 32)])
    for p in range(mc_n):
        orc.head_pass(feat, idx[:feat.shape[0]], p, seed)
    dt_head = time.time() - t0
    return {'value': full, 'unit': 'tiles/s', 'cores': threads, 'kind': 'port',
            'sample': f'{passes} complete fp32 forward passes (PyTorch-CPU oracle, dropout on) over a batch of '
                      f'{batch} synthetic 299x299x3 tiles ({passes * batch} tile-passes) in {dt_full:.1f} s, scaled to '
                      f'MC={mc_n} passes per tile',
            'head_mode_value': feat.shape[0] / dt_head, 'cpu': _cpu_name(), 'os_cpu_count': os.cpu_count()}


# ----------------------------------------------------------------------------- the benchmark proper
def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from biscuit_amd import distributed as D
    from biscuit_amd.engine import EnginePool
    from biscuit_amd.weights import synthetic_weights

    rank, world, local = D.init_from_env('cuda', backend=args.dist_backend, local_device=args.local_device)
    if world != args.gpus:
        raise SystemExit(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus} '
                         f'(or without torchrun: bench.py starts its own ranks)')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    # this rank's share of the node's cores, next to its GPU where sysfs tells -- before any thread pool exists
    rank_cpus = D.pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)), device_index=local)
    # collectives run on the GPU (RCCL) unless a CPU backend was forced for single-GPU testing
    coll_dev = dev if (world == 1 or dist.get_backend() == 'nccl') else torch.device('cpu')

    # proof that the collective backend really spans the ranks: a sum of ones over it
    coll_ranks = world
    if world > 1:
        one = torch.ones(1, dtype=torch.int32, device=coll_dev)
        dist.all_reduce(one)
        coll_ranks = int(one.item())
    # rccl_ranks = ranks an RCCL communicator really spanned in this run: 0 when no process group exists (N = 1) or the
    # group is a CPU backend
    coll = {'backend': dist.get_backend() if world > 1 else None, 'ranks_seen': coll_ranks,
            'rccl_ranks': coll_ranks if (world > 1 and dist.get_backend() == 'nccl') else 0}

    weights = synthetic_weights(1)
    pool_e = EnginePool(weights, n_streams=args.streams, dtype=args.dtype, max_batch=args.batch, max_mc=args.mc,
                        device=local, size_grids=args.size_grids)
    eng = pool_e.engines[0]
    NS = len(pool_e)
    B, K, Wm = args.batch, args.steps, args.warmup
    seed = 1234

    # synthetic tiles, generated on the device: 4 batches resident, cycled
    # (config 3 is ONE dataset sharded over the ranks: every rank generates the same tiles; config 2 is per-rank work)
    g = torch.Generator(device=dev).manual_seed(100 + (rank if args.workload == 'cfg2' else 0))
    pool = [torch.randint(0, 256, (B, 299, 299, 3), dtype=torch.uint8, device=dev, generator=g)
            for _ in range(4)]
    scratch = [torch.empty_like(pool[0]) for _ in range(NS)]      # stain-normalised copy, one per stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.workload == 'cfg3':
        out = run_cfg3(args, rank, world, dev, pool_e, pool, barrier, max_over_ranks)
        out['collective'] = coll
        out['rccl_ranks'] = coll['rccl_ranks']
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    NSTEP = max(K, Wm, 8)                         # the per-kernel pass and the full-mode leg run up to 8 steps
    n_slides_local = (NSTEP * B + TILES_PER_SLIDE - 1) // TILES_PER_SLIDE + 1
    slide_of = [torch.div(torch.arange(s * B, (s + 1) * B, device=dev), TILES_PER_SLIDE,
                          rounding_mode='floor').to(torch.int32) for s in range(NSTEP)]
    NB = max(NS, 2)                               # result buffers / accumulators: one per context of either pool
    mean = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(NB)]
    std = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(NB)]
    tile_base = rank * K * B                      # global tile index of this rank's shard

    def zero_acc(n=NB):
        # one fixed-point accumulator triple per stream; integer sums add exactly at the end
        return [(torch.zeros(n_slides_local, dtype=torch.int64, device=dev),
                 torch.zeros(n_slides_local, dtype=torch.int64, device=dev),
                 torch.zeros(n_slides_local, dtype=torch.int32, device=dev)) for _ in range(n)]

    def step(pe, i, acc, mode, stain=False):
        k = i % len(pe)               # batches in flight (set by the calibration below)

        def work(e):
            src = pool[i % 4]
            if stain:
                src = e.reinhard_fast(src, NORM_FIT['target_means'], NORM_FIT['target_stds'], out=scratch[k])
            e.mc_infer(src, args.mc, seed, tile_idx0=tile_base + i * B, mc_mode=mode, out=(mean[k], std[k]))
            e.slide_reduce(mean[k], std[k], slide_of[i], n_slides_local, acc=acc[k])
        pe.run(k, work)

    def finish(pe, acc):
        torch.cuda.synchronize()
        tot = tuple(sum(a[j] for a in acc[1:]) + acc[0][j] if len(acc) > 1 else acc[0][j] for j in range(3))
        return pe.engines[0].slide_finish(tot)

    def timed(mode, steps, pe=None, stain=False):
        pe = pe or pool_e
        acc = zero_acc()
        for i in range(Wm):
            step(pe, i, acc, mode, stain)
        acc = zero_acc()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(pe, i, acc, mode, stain)
        mp, mu, cnt = finish(pe, acc)
        if world > 1:                              # the path's one collective (RCCL over xGMI)
            ids = np.arange(rank * n_slides_local, (rank + 1) * n_slides_local)
            D.gather_slide_results(ids, mp.cpu().numpy(), mu.cpu().numpy(), cnt.cpu().numpy(),
                                   world * n_slides_local, n_slides_local, device=coll_dev)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert int(cnt.sum()) == steps * B and bool(torch.isfinite(mp[cnt > 0]).all())
        return dt

    # Untimed calibration of the number of batches in flight.  Each batch in flight owns 1/n of the chip's
    # XCDs (CU-masked streams), which lets batches run out of phase (one's HBM-bound prologues and store
    # drains under another's compute).  But n in flight only pays when the K timed steps fill whole rounds of n
    # (10 steps on 4 quarter-chips are 3 rounds), and plain streams (if CU masks are unavailable) are bimodal.
    # So time min(K, 32) steps each way and keep the fastest; every rank adopts the same choice.
    cands = sorted({n for n in (NS, NS // 2, 1) if n >= 1}, reverse=True)
    if len(cands) > 1 and not args.fixed_streams:
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
                               f'Xception {args.dtype} storage + MFMA, fp32 accumulation / BN / MC head, MC={args.mc}, '
                               f'batch={B}, {K * B} tiles/GPU resident in HBM'
                               + ('; f16 is the 16-bit mode that holds the 1e-3 tile/slide tolerance on O(1)-logit weights '
                                  '(bf16, the type config 2 names, runs 5-6 % slower and misses it: bf16_value)'
                                  if args.dtype == 'f16' else ''),
                   'mc_mode': args.mode, 'mc_n': args.mc, 'batch': B, 'hip_streams': streams_used,
                   'grids_sized_for_mask': bool(args.size_grids),
                   'parallelism': f'slide-sharded dp{world}, one all-gather of slide (pred, sigma, n)'},
        'rccl_ranks': coll['rccl_ranks'], 'collective': coll, 'host_cores_of_rank0': len(rank_cpus),
    }

    flop_head = FLOP_BACKBONE + args.mc * FLOP_HEAD_PASS
    flop_full = args.mc * (FLOP_BACKBONE + FLOP_HEAD_PASS)
    if rank == 0:
        flop_tile = flop_head if args.mode == 'head' else flop_full
        per_gpu = value / world
        out['path_roofline'] = {
            'hbm_frac': per_gpu * BYTES_PER_TILE_BF16 / PEAK_HBM if args.dtype in HALF else None,
            'hbm_frac_vs_measured_peak': per_gpu * BYTES_PER_TILE_BF16 / PEAK_HBM_MEASURED if args.dtype in HALF else None,
            'mfma_frac': per_gpu * flop_tile / (PEAK_BF16 if args.dtype in HALF else PEAK_F32),
            'bytes_per_tile': BYTES_PER_TILE_BF16, 'flop_per_tile': flop_tile}

    solo = world == 1 and rank == 0
    leg_s = {}                                          # wall seconds of the extra legs (what a default run spends where)
    t_leg = time.perf_counter()

    def leg_done(name):
        nonlocal t_leg
        now = time.perf_counter()
        leg_s[name] = round(now - t_leg, 1)
        t_leg = now
    # the other MC structure on the same kernels (N=1 only; a few steps)
    if solo:
        other = 'full' if args.mode == 'head' else 'head'
        k2 = max(2, len(pool_e)) if other == 'full' else K     # a batch per stream keeps every XCD group busy
        dt2 = timed(other, k2)
        out[f'{other}_mode_value'] = k2 * B / dt2
    # hp.py:19's stain normaliser inside the timed region (results.py:251-252 applies it per tile)
    if solo and not args.no_extras:
        dt3 = timed(args.mode, K, stain=True)
        out['with_reinhard_value'] = K * B / dt3
        out['with_reinhard_ms_per_step'] = dt3 / K * 1e3
    leg_done('other_mode_and_stain')

    # per-kernel roofline: HIP events on the launch stream around every launch
    if not args.no_profile and rank == 0:
        acc = zero_acc()
        torch.cuda.synchronize()
        eng.profile_enable(True)
        psteps = 4
        # Engine 0 on a plain (whole-chip) stream of its own, nothing else in flight: the events bracket
        # each launch on the stream it runs on and give the kernel's own duration, the figure the
        # single-stream rocprofv3 trace under profiles/ is comparable with.  (In the timed region each
        # batch owns a share of the chip and several run side by side; see `in_situ` below.)
        solo_s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(solo_s):
            for i in range(psteps):
                eng.mc_infer(pool[i % 4], args.mc, seed, tile_idx0=tile_base + i * B, mc_mode=args.mode,
                             out=(mean[0], std[0]))
                eng.slide_reduce(mean[0], std[0], slide_of[i], n_slides_local, acc=acc[0])
        solo_s.synchronize()
        # leaf entries only: a `split_*` class brackets its two children (dw3x3_* + gemm_*), which have their own entries
        ents = [e for e in eng.profile_read() if not e.name.startswith('split_')]
        eng.profile_enable(False)
        tot = sum(e.ms for e in ents)
        ents.sort(key=lambda e: -e.ms)
        dom = ents[0]
        avg_s = dom.ms / dom.launches * 1e-3
        es = 2 if args.dtype in HALF else 4
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
        # HBM bytes per launch: NOT measured by this run (hardware counters need rocprofv3 around the process); the
        # figure is the one the committed PMC passes produced for this kernel, dtype and batch, and says so
        traffic, traffic_source = None, 'not measured in this run; no committed PMC figure for this kernel / dtype / batch'
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            ent = tj.get(args.dtype, {}).get(dom.name) if B == 256 else None
            if ent:
                traffic = ent['corrected_bytes_per_launch']
                traffic_source = ('profiles/traffic.json (committed, not measured in this run): ' + tj.get('_source', ''))
        except (OSError, ValueError, KeyError, AttributeError):
            pass
        # the same kernel class inside the timed region: its work per step over its share of the step time
        in_situ = (dom.flops if bound == 'mfma' else dom.bytes) * (dom.launches / psteps) / \
                  ((dom.ms / tot) * (dt / K)) / (1e12 if bound == 'mfma' else 1e9)
        out['roofline'] = {'kernel': dom.name, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit,
                           'frac': ach / peak, 'traffic': traffic, 'traffic_source': traffic_source,
                           'in_situ': {'achieved': in_situ, 'frac': in_situ / peak,
                                       'note': 'work per step / (share of kernel time x measured step time), '
                                               f'{streams_used} batches in flight on disjoint XCD groups'},
                           'launches_per_step': dom.launches / psteps, 'avg_launch_ms': dom.ms / dom.launches,
                           'share_of_step': dom.ms / tot,
                           'algorithmic_flops_per_launch': dom.flops, 'algorithmic_bytes_per_launch': dom.bytes}
        pk_f = (PEAK_BF16 if es == 2 else PEAK_F32) / 1e12

        def kern(e):
            tf = e.flops / (e.ms / e.launches * 1e-3) / 1e12
            gb = e.bytes / (e.ms / e.launches * 1e-3) / 1e9
            return {'name': e.name, 'launches_per_step': e.launches / psteps, 'ms_per_launch': e.ms / e.launches,
                    'share': e.ms / tot, 'tflops': tf, 'gbps': gb,
                    # fraction of the roofline that binds the launch: the larger of its MFMA and HBM fractions
                    'frac_of_bound': max(tf / pk_f, gb / (PEAK_HBM / 1e9))}
        out['kernels'] = [kern(e) for e in ents[:48]]
        # the HBM-bound entry side of the network (staging, stem, blocks 1-3): ms per step and the same bytes at the
        # measured copy rate
        entry = [e for e in ents if e.name.startswith(ENTRY_SIDE)]
        out['entry_side_ms'] = sum(e.ms for e in entry) / psteps
        out['entry_side_floor_ms'] = sum(e.bytes * e.launches for e in entry) / psteps / PEAK_HBM_MEASURED * 1e3
        out['entry_side_kernels'] = sorted(e.name for e in entry)

    leg_done('kernel_profile')
    if solo and not args.no_extras:
        out['b1_latency'] = b1_latency(eng, args.mc)
        pool_e.synchronize()
        nfl = len(pool_e)
        for other_dt in [d for d in ('f16', 'bf16', 'f32') if d != args.dtype]:
            # the other storage types through the same loop (16-bit: the same number of batches in flight as the headline)
            pe2 = EnginePool(weights, n_streams=1 if other_dt == 'f32' else nfl, dtype=other_dt, max_batch=B, max_mc=args.mc,
                             device=local)
            k4 = 4 if other_dt == 'f32' else K
            dt4 = timed(args.mode, k4, pe=pe2)
            out[f'{other_dt}_value'] = k4 * B / dt4
            out[f'{other_dt}_ms_per_step'] = dt4 / k4 * 1e3
            out[f'{other_dt}_mfma_frac'] = out[f'{other_dt}_value'] * flop_head / (PEAK_F32 if other_dt == 'f32' else PEAK_BF16)
            pe2.synchronize()
            pe2.close()
            del pe2
        leg_done('other_dtypes')
        try:
            out['tfrecords'] = tfrecord_leg(pool_e, args)
        except Exception as e:                                  # an extra leg never takes the headline down
            out['tfrecords'] = {'error': f'{type(e).__name__}: {e}'}
        leg_done('tfrecords')
        try:
            out['host_tiles'] = host_tiles_leg(pool_e, args)
        except Exception as e:
            out['host_tiles'] = {'error': f'{type(e).__name__}: {e}'}
        leg_done('host_tiles')

    if solo and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(weights, args.mc, seed, args.cpu_tiles, args.cpu_budget)
        out['speedup_vs_cpu_full'] = (out.get('full_mode_value') or value) / out['cpu_baseline']['value']
        out['speedup_headline_vs_cpu_full'] = value / out['cpu_baseline']['value']
        leg_done('cpu_baseline')
    if leg_s:
        out['leg_seconds'] = leg_s

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_cfg3(args, rank, world, dev, pool_e, pool, barrier, max_over_ranks):
    """BASELINE.json config 3: S slides x T tiles, LPT-sharded, through ``inference.evaluate``; strong scaling."""
    import numpy as np
    import torch
    from biscuit_amd import distributed as D
    from biscuit_amd.inference import Slide, evaluate
    S, T, B = args.slides, args.tiles_per_slide, args.batch
    allt = torch.cat(pool)                                       # the resident synthetic tiles, cycled
    npool = allt.shape[0]

    def tiles_of(i, count):
        def load():
            if count <= npool:                                   # a window of the resident tiles: no copy (a gather of 1 000 tiles
                start = (i * T) % (npool - count + 1)            # through index_select took 0.9 ms, 2 % of config 3's run)
                return allt[start:start + count]
            idx = (torch.arange(count, device=dev) + i * T) % npool
            return allt.index_select(0, idx)
        return load
    counts = [T - (7 * i) % max(T // 2, 1) for i in range(S)] if args.ragged else [T] * S
    slides = [Slide(f's{i:05d}', tiles_of(i, c), c, y_true=i % 2) for i, c in enumerate(counts)]
    NT = sum(counts)
    wt = min(T, 2 * B)
    warm = [Slide(f'w{i}', tiles_of(i, wt), wt, y_true=0) for i in range(world * max(4, args.warmup))]
    # batches in flight: calibrated like config 2's loop -- the warm-up slides once with two batches on half the chip each,
    # once with one on the whole chip, the faster setting kept (every rank adopts the same choice)
    evaluate(pool_e, warm, mc_n=args.mc, seed=1234, batch=B, mc_mode=args.mode, keep_tiles=False, rank=rank, world=world)
    cands = [n for n in (2, 1) if n <= len(pool_e.engines)]
    times = []
    for nfl in cands:
        pool_e.set_in_flight(nfl)
        barrier()
        t0 = time.perf_counter()
        evaluate(pool_e, warm, mc_n=args.mc, seed=1234, batch=B, mc_mode=args.mode, keep_tiles=False, rank=rank, world=world)
        barrier()
        times.append(max_over_ranks(time.perf_counter() - t0))
    pool_e.set_in_flight(cands[int(np.argmin(times))])
    barrier()
    t0 = time.perf_counter()
    res = evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=B, mc_mode=args.mode, keep_tiles=False,
                   rank=rank, world=world)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    assert int(res.slide_count.sum()) == NT and np.isfinite(res.slide_pred).all()
    # what the f16 range monitor costs (evaluate's default: the eight taps on eight tiles every 200 batches, no host sync): the same run
    # without it
    barrier()
    t0 = time.perf_counter()
    evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=B, mc_mode=args.mode, keep_tiles=False, rank=rank, world=world,
             headroom_every=0)
    barrier()
    dt_nomon = max_over_ranks(time.perf_counter() - t0)
    # the same run WITH the path's product: the tile table streamed to disk while the GPU works (every rank its shard, closed before
    # the all-gather; rank 0 splices them into tile_predictions_eval.csv) -- rows written and the file closed inside the timed region
    table = None
    if not args.no_table:
        import shutil
        import tempfile
        # one directory all ranks of this run derive by themselves (no object collective): the rendezvous port names the run
        tdir = os.path.join(tempfile.gettempdir(), f"bq_table_{os.environ.get('MASTER_PORT', 'p')}_{os.getppid() if world > 1 else os.getpid()}")
        os.makedirs(tdir, exist_ok=True)
        try:
            barrier()
            t0 = time.perf_counter()
            rt = evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=B, mc_mode=args.mode, keep_tiles=False,
                          rank=rank, world=world, save_dir=tdir)
            barrier()
            dtt = max_over_ranks(time.perf_counter() - t0)
            assert np.array_equal(rt.slide_pred, res.slide_pred) and np.array_equal(rt.slide_unc, res.slide_unc)
            if rank == 0:
                nbytes = os.path.getsize(rt.table_path)
                with open(rt.table_path, 'rb') as f:
                    nrows = sum(chunk.count(b'\n') for chunk in iter(lambda: f.read(1 << 24), b'')) - 1
                assert nrows == NT, (nrows, NT)
                import hashlib
                hs = hashlib.sha256()
                with open(rt.table_path, 'rb') as f:
                    for chunk in iter(lambda: f.read(1 << 24), b''):
                        hs.update(chunk)
                table = {'sha256': hs.hexdigest(), 'with_table_value': NT / dtt, 'seconds': dtt, 'rows': nrows, 'bytes': nbytes,
                         'ratio_to_value': dt / dtt, 'file': os.path.basename(rt.table_path),
                         'note': 'evaluate(save_dir=...): rows formatted and written by libbiscuit_io on a host thread while the GPU '
                                 'works; at N > 1 per-rank shards closed before the all-gather, spliced by rank 0 -- all inside the time'}
        finally:
            barrier()
            if rank == 0:
                shutil.rmtree(tdir, ignore_errors=True)
    parts = D.partition_slides(counts, world)
    steps = max(-(-sum(counts[i] for i in p) // B) for p in parts)
    return {'metric': 'tiles/sec at MC-dropout=30, 299x299x3 (Xception, slide-level pred/sigma reduce)',
            'value': NT / dt, 'unit': 'tiles/s', 'n_gpus': world, 'steps': steps, 'warmup': args.warmup,
            'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic', 'seconds': dt,
            'with_table_value': table['with_table_value'] if table else None, 'tile_table': table,
            'range_monitor': {'checks': int(res.f16_checks), 'min_headroom': (None if not np.isfinite(res.f16_headroom) else float(res.f16_headroom)),
                              'value_without': NT / dt_nomon, 'cost_frac': dt / dt_nomon - 1.0},
            # the slide table every rank holds after the gather, as a digest: equal at every world size (Philox counters are global
            # tile indices, the slide sums order-free fixed point)
            'slide_table_sha256': __import__('hashlib').sha256(np.ascontiguousarray(res.slide_pred).tobytes()
                                                               + np.ascontiguousarray(res.slide_unc).tobytes()).hexdigest(),
            'config': {'workload': f'BASELINE.json config 3: {S} synthetic slides x {T} tiles (299x299x3), LPT-sharded over '
                                   f'{world} GPU(s), Xception {args.dtype} + fp32 MC head, MC={args.mc}, batch={B}, through '
                                   f'biscuit_amd.inference.evaluate', 'mc_mode': args.mode, 'mc_n': args.mc, 'batch': B,
                       'slides_per_rank': [len(p) for p in parts], 'tiles_per_rank': [sum(counts[i] for i in p) for p in parts],
                       'ragged': bool(args.ragged), 'hip_streams': len(pool_e),
                       'parallelism': f'slide-sharded dp{world}, one all-gather of slide (pred, sigma, n)'},
            'path_roofline': {'hbm_frac': NT / dt / world * BYTES_PER_TILE_BF16 / PEAK_HBM if args.dtype in HALF else None,
                              'mfma_frac': NT / dt / world * (FLOP_BACKBONE + args.mc * FLOP_HEAD_PASS) /
                              (PEAK_BF16 if args.dtype in HALF else PEAK_F32)}}


def b1_latency(eng, mc_n, calls=50):
    """results.py:250-258 as the heatmap loop runs it: one standardised 299x299x3 float tile per call,
    ``UncertaintyInterface(batch) -> (mean, std)`` with MC passes folded in the head; host wall time per
    call (H2D of the tile, ~60 launches, D2H of 4 floats) and the device time between HIP events."""
    import numpy as np
    import torch
    from biscuit_amd.engine import UncertaintyInterface
    itf = UncertaintyInterface(eng, uq_n=mc_n, seed=5)
    x = np.random.default_rng(0).normal(0, 1, (1, 299, 299, 3)).astype(np.float32)
    for _ in range(5):
        itf(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        itf(x)
    wall = (time.perf_counter() - t0) / calls
    xd = torch.from_numpy(x).to(eng.device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(calls):
        itf.device_call(xd)
    b.record()
    b.synchronize()
    res = {'us_per_call_host_wall': wall * 1e6, 'us_per_call_device': a.elapsed_time(b) / calls * 1e3, 'mc_n': mc_n,
           'calls': calls, 'graph': False}
    if hasattr(itf, 'enable_graph'):
        try:
            itf.enable_graph()
            for _ in range(3):
                itf.device_call(xd)
            a.record()
            for _ in range(calls):
                itf.device_call(xd)
            b.record()
            b.synchronize()
            res['us_per_call_device_graph'] = a.elapsed_time(b) / calls * 1e3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                itf(x)
            res['us_per_call_host_wall_graph'] = (time.perf_counter() - t0) / calls * 1e6
            res['graph'] = True
        except Exception as e:
            res['graph_error'] = f'{type(e).__name__}: {e}'
    return res


class _HostTiles:
    """A slide whose decoded uint8 tiles already sit in host memory, as a chunk source for ``evaluate``'s pinned ring: ``read``
    copies tiles [first, first + count) into the ring's buffer -- the boundary case "the caller hands over host buffers"."""
    rows = False

    def __init__(self, tiles):
        self.tiles = tiles
        self.tile_px = int(tiles.shape[1])      # (the pinned ring is sized per tile size, for either chunk layout)

    def chunk_shape(self, count):
        return (count,) + tuple(self.tiles.shape[1:])

    def read(self, first, count, out):
        out[...] = self.tiles[first:first + count]

    def close(self):
        pass


def host_tiles_leg(pool_e, args, n_slides=8, tiles_per_slide=1000):
    """The PCIe-inclusive rate: ``evaluate`` over slides whose decoded tiles are in (pageable) HOST memory -- host copy into the
    pinned ring, H2D on the copy stream, the same kernels.  No decode: what is left of the TFRecord leg when the host is not the
    bound.  Never the headline (inputs of `value` are resident in HBM)."""
    import numpy as np
    from biscuit_amd.inference import Slide, evaluate
    from biscuit_amd.synthetic import make_tiles
    base = make_tiles(250, seed=33)                                             # 67 MB of distinct tiles, repeated per slide
    slides = []
    for s in range(n_slides):
        t = np.ascontiguousarray(np.concatenate([np.roll(base, s, axis=0)] * (tiles_per_slide // 250))[:tiles_per_slide])
        slides.append(Slide(name=f'h{s}', tiles=t, n_tiles=len(t), y_true=s % 2, source=_HostTiles(t)))
    n = sum(s.n_tiles for s in slides)
    evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)        # warm-up (ring allocation)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)
        ts.append(time.perf_counter() - t0)
        assert int(res.slide_count.sum()) == n
    dt = sorted(ts)[1]
    return {'tiles': n, 'value': n / dt, 'unit': 'tiles/s', 'runs': [n / t for t in ts], 'bytes_per_tile': int(np.prod(base.shape[1:])),
            'h2d_gbps': n * float(np.prod(base.shape[1:])) / dt / 1e9,
            'note': 'evaluate() over 8 slides x 1000 decoded uint8 tiles in pageable host memory: host copy into the ring of three '
                    'pinned 512-tile buffers, H2D on its own stream, kernels; median of three runs'}


def tfrecord_leg(pool_e, args, n_slides=32, tiles_per_slide=1024, n_files=8):
    """``evaluate`` fed from self-written PNG TFRecords (configure.py:118-124: PNG tiles, one file per slide):
    host decode (libbiscuit_io.so on the box's cores) -> pinned ring -> H2D -> the same kernels.  Two kinds of tiles: 'noise'
    (make_tiles' default grain: nearly incompressible PNGs, the figures of rounds 2-4) and 'photo' (smooth texture + grain: 155 KB
    per tile, the size of a real H&E tile).  Per kind 32 slides x 1 024 tiles = 32 768 tiles per run (8 files on disk, the other 24 slides are links
    to them: round 5's first runs of 8 192 tiles spent 8 % of their 0.28 s filling and draining the pipeline); ``decode_only`` is the WARM rate of the decoder alone
    (second pass over the same files: page cache and thread pool warm) -- the ceiling the end-to-end rates are to be read against
    (round 4 reported one cold pass over 2 048 tiles, which came out BELOW the end-to-end rates it was meant to bound)."""
    import shutil
    import tempfile
    import numpy as np
    from biscuit_amd import tfrecord, tfrecord_native
    from biscuit_amd.inference import evaluate, pick_unfilter_mode, slides_from_tfrecords
    from biscuit_amd.synthetic import make_tiles
    tfrecord_native.load()
    n = n_slides * tiles_per_slide
    lab = {f's{s}': s % 2 for s in range(n_slides)}

    def one_kind(d, grain):
        base = [tfrecord.encode_image(t) for t in make_tiles(32, seed=21, grain=grain)]      # 32 distinct PNGs, written many times
        paths = []
        for s in range(n_slides):
            p = os.path.join(d, f's{s}.tfrecords')
            if s < n_files:
                tfrecord.write_slide(p, f's{s}', [base[(i + s) % 32] for i in range(tiles_per_slide)],
                                     np.zeros((tiles_per_slide, 2), np.int64))
            else:
                os.symlink(os.path.join(d, f's{s % n_files}.tfrecords'), p)
            paths.append(p)
        nbytes = sum(os.path.getsize(p) for p in paths)      # (getsize follows the links: the bytes a run reads)
        # the decoder alone, the way evaluate() drives it: 512-tile chunks into ONE reused buffer (a fresh 275 MB array per slide
        # would time the page faults of its first touch, not the decoder) -- cold, then warm twice; then the same stopping at the
        # filtered scanlines (what the GPU un-filter mode asks of the host)
        from biscuit_amd.inference import CHUNK_TILES, TFRecordSource
        def decode_pass(rows):
            buf = np.empty((CHUNK_TILES, 299, 1 + 3 * 299) if rows else (CHUNK_TILES, 299, 299, 3), np.uint8)
            t0 = time.perf_counter()
            for p in paths:
                src = TFRecordSource(p, tiles_per_slide, 299, rows=rows)
                for first in range(0, tiles_per_slide, CHUNK_TILES):
                    cnt = min(CHUNK_TILES, tiles_per_slide - first)
                    src.read(first, cnt, buf[:cnt])
                src.close()
            return time.perf_counter() - t0
        dec = [decode_pass(False) for _ in range(3)]
        rows_dec = [decode_pass(True) for _ in range(2)]
        slides = slides_from_tfrecords(paths, lab)
        gslides = slides_from_tfrecords(paths, lab, gpu_unfilter=True)
        auto, host_rate, rows_rate = pick_unfilter_mode(paths[0])
        # warm-up in both modes, then the two modes ALTERNATE (A/B/A/B/A/B) and the medians are reported: whichever mode runs
        # first on a box otherwise looks slower (round 3 read +23 % into that once)
        evaluate(pool_e, slides[:2], mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)
        evaluate(pool_e, gslides[:2], mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)
        th, tg, tt = [], [], []
        for rep in range(3):
            t0 = time.perf_counter()
            res = evaluate(pool_e, slides, mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)
            th.append(time.perf_counter() - t0)
            assert int(res.slide_count.sum()) == n
            t0 = time.perf_counter()
            evaluate(pool_e, gslides, mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False)
            tg.append(time.perf_counter() - t0)
            # the same run with the path's product: the tile table streamed to disk, file closed inside the time
            t0 = time.perf_counter()
            rt = evaluate(pool_e, gslides if auto else slides, mc_n=args.mc, seed=1234, batch=args.batch, keep_tiles=False,
                          save_dir=os.path.join(d, 'table'))
            tt.append(time.perf_counter() - t0)
            assert rt.table_rows == n
        dt, gdt, tdt = sorted(th)[1], sorted(tg)[1], sorted(tt)[1]
        return {'tiles': n, 'value': n / dt, 'unit': 'tiles/s', 'gpu_unfilter_value': n / gdt,
                'with_table_value': n / tdt, 'with_table_ratio': (gdt if auto else dt) / tdt, 'runs_with_table': [n / t for t in tt],
                'decode_only_tiles_per_s': n / min(dec[1:]), 'decode_only_cold_tiles_per_s': n / dec[0],
                'decode_rows_only_tiles_per_s': n / min(rows_dec),
                'runs_host_unfilter': [n / t for t in th], 'runs_gpu_unfilter': [n / t for t in tg],
                'auto_mode': 'gpu_unfilter' if auto else 'host', 'auto_probe_tiles_per_s': {'host': host_rate, 'rows': rows_rate},
                'png_bytes_per_tile': nbytes / n}

    d = tempfile.mkdtemp(prefix='bq_tfr_')
    try:
        os.makedirs(os.path.join(d, 'noise'))
        out = one_kind(os.path.join(d, 'noise'), 18.0)
        shutil.rmtree(os.path.join(d, 'noise'), ignore_errors=True)
        os.makedirs(os.path.join(d, 'photo'))
        out['photo'] = one_kind(os.path.join(d, 'photo'), 4.0)
        shutil.rmtree(os.path.join(d, 'photo'), ignore_errors=True)
        # the same tiles as JPEG records (Slideflow's img_format='jpg'): decode only, one file
        jbase = [tfrecord.encode_image(t, 'JPEG') for t in make_tiles(32, seed=21)]
        jp = os.path.join(d, 'j.tfrecords')
        tfrecord.write_slide(jp, 'j', [jbase[i % 32] for i in range(tiles_per_slide)], np.zeros((tiles_per_slide, 2), np.int64))
        tfrecord.read_slide(jp, 299)
        t0 = time.perf_counter()
        for _ in range(4):
            tfrecord.read_slide(jp, 299)
        jdec = (time.perf_counter() - t0) / 4
        out.update({'jpeg_decode_only_tiles_per_s': tiles_per_slide / jdec, 'jpeg_bytes_per_tile': os.path.getsize(jp) / tiles_per_slide,
                    'host_cores': usable_cores(), 'decoder_threads': tfrecord_native.default_threads(),
                    'note': 'end to end from PNG TFRecords incl. host decode, H2D, kernels, tile table: 32 slides x 1 024 tiles per run, '
                            'medians of three alternated runs per mode; 512-tile chunks through a ring of three pinned buffers, copies on '
                            'their own stream; decode_only = the decoder alone, warm (best of two passes after a cold one): the host '
                            'bound of the end-to-end rates; top level: nearly incompressible tiles, `photo`: photo-like ones'})
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def selftest_rank(args):
    """A rank of the launcher's self-test (tests/test_distributed.py; no GPU is touched): rank R of ``--selftest-exit R:CODE`` leaves
    with CODE before the rendezvous, every other rank goes through the rendezvous and one barrier of a gloo group."""
    r, code = (-1, 0) if args.selftest_exit == 'none' else (int(x) for x in args.selftest_exit.split(':'))
    if int(os.environ.get('RANK', '0')) == r:
        sys.exit(code)
    import torch.distributed as dist
    from biscuit_amd import distributed as D
    rank, world, _ = D.init_from_env('cpu', backend='gloo')
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({'selftest': 'ok', 'world': world}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.selftest_exit:
        return selftest_rank(args)
    run(args)


if __name__ == '__main__':
    main()
